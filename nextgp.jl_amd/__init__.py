"""nextgp.jl_amd -- MI355X-native marker-effect Gibbs sampler behind NextGP.jl's runLMEM interface.

The directory name contains a dot, so it is loaded by path (see `ngp_pkg.py` at the repo root):
    from ngp_pkg import load_pkg; ngp = load_pkg()
"""
from ._lib import (LIB_PATH, METHOD_BAYESB, METHOD_BAYESC, METHOD_BAYESPR, METHOD_BAYESR, METHOD_TUPLE, SYMBOLS, NextGPHipError, Sampler, load,  # noqa: F401
                   read_panel_header, read_sample_file, tuple_columns, tuple_panel, tuple_span, write_panel_file)
from .api import (BayesB, BayesC, BayesPR, BayesR, Random, SNP, design_columns, is_panel_file, parse_formula, prep2RegionData,
                  read_genotypes, read_panel_file, runLMEM, samples_to_out_files, summaryMCMC)  # noqa: F401,E402
from . import multichain  # noqa: F401,E402
