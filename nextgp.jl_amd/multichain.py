"""Independent chains, one per GPU/rank: packing and the single end-of-run reduction of posterior sums.

The path shards across chains only: no collective inside the sweep; after the last
iteration every rank holds a packed vector [sum_beta P | sum_beta2 P | sum_delta P | sum_varBeta nvb |
sum_pi 2*nsets | BayesR class sums | fixed-effect sums | sum_varE | sum_b | nKept] (ngp_export_posterior_device) and ONE all-reduce(sum) over
RCCL (backend "nccl" on ROCm; "gloo" in CPU tests) yields the pooled sums."""
import numpy as np


def posterior_len(P, nvb, nsets, nclasses=0, nfix=0):
    """nclasses = sum of K over the BayesR sets (their class-probability sums travel too); nfix = fixed-effect columns beyond
    the intercept (all sets)."""
    return 3 * P + nvb + 2 * nsets + nclasses + nfix + 3


def pack_posterior(ps, P, nvb, nsets, class_sums=(), fixed_sums=()):
    """dict from Sampler.get_posterior_sums() (or the oracle) -> packed float64 vector (host); class_sums = the BayesR sets'
    class-probability sums, concatenated set by set; fixed_sums = Sampler.get_fixed()["sum_b"]."""
    class_sums = np.asarray(class_sums, dtype=np.float64)
    fixed_sums = np.asarray(fixed_sums, dtype=np.float64)
    out = np.empty(posterior_len(P, nvb, nsets, len(class_sums), len(fixed_sums)))
    o = 3 * P + nvb + 2 * nsets
    out[o:o + len(class_sums)] = class_sums
    out[o + len(class_sums):o + len(class_sums) + len(fixed_sums)] = fixed_sums
    out[0:P] = ps["sum_beta"]; out[P:2 * P] = ps["sum_beta2"]; out[2 * P:3 * P] = ps["sum_delta"]
    out[3 * P:3 * P + nvb] = ps["sum_varBeta"]
    out[3 * P + nvb:3 * P + nvb + 2 * nsets] = ps["sum_pi"]
    out[-3], out[-2], out[-1] = ps["sum_varE"], ps["sum_b"], ps["nKept"]
    return out


def unpack_means(buf, P, nvb, nsets, nclasses=0, nfix=0):
    """packed (possibly all-reduced) sums -> posterior means over all kept samples of all chains."""
    buf = np.asarray(buf, dtype=np.float64)
    n = max(buf[-1], 1.0)
    mean = buf[0:P] / n
    o = 3 * P + nvb + 2 * nsets
    return dict(nKept=int(round(buf[-1])), beta=mean, beta_sd=np.sqrt(np.maximum(buf[P:2 * P] / n - mean ** 2, 0.0)),
                delta=buf[2 * P:3 * P] / n, varBeta=buf[3 * P:3 * P + nvb] / n,
                pi=buf[3 * P + nvb:3 * P + nvb + 2 * nsets] / n,
                class_pi=buf[o:o + nclasses] / n, b_fixed=buf[o + nclasses:o + nclasses + nfix] / n, varE=buf[-3] / n, b=buf[-2] / n)


def allreduce_posterior(tensor, group=None):
    """In-place sum over ranks of the packed tensor (device tensor on GPUs, CPU tensor under gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor
