# Optional statistical cross-check against the real reference, for a box that has Julia + NextGP.jl installed
# (SURVEY.md section 8c: bit parity with Julia's RNG is unattainable; posterior means must agree within Monte-Carlo error).
#
#   julia tools/compare_with_julia.jl <genotypes.txt> <phenotypes.txt> [nChain nBurn nThin]
#
# Runs the same BayesPR model twice from the SAME prepared model terms: once through the reference sampler
# (src/samplers.jl:23) and once through the coarse seam of libnextgp_hip (nextgp.jl_amd/julia/NextGPHIP.jl), and prints the
# correlation of the posterior mean marker effects and the relative difference of the posterior mean residual variance.
# Not used by any test here: the build image has no Julia.
using NextGP, DataFrames, DelimitedFiles, Statistics

geno, pheno = ARGS[1], ARGS[2]
nChain = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 2000
nBurn  = length(ARGS) >= 4 ? parse(Int, ARGS[4]) : 500
nThin  = length(ARGS) >= 5 ? parse(Int, ARGS[5]) : 5
y = vec(readdlm(pheno))
data = DataFrame(y = y)
v = 0.5 * var(y) / size(readdlm(geno), 2)
f = @formula(y ~ 1 + SNP(M, geno))
priors = Dict(:M => BayesPR(9999, v), :e => Random("I", 0.5 * var(y)))

refdir, gpudir = mktempdir(), mktempdir()
# needs the one-line switch of INTEGRATION.md in src/MCMC.jl:39 (it looks at ENV["NEXTGP_HIP"])
ENV["NEXTGP_HIP"] = "0"; runLMEM(f, data, nChain, nBurn, nThin; VCV = priors, outFolder = refdir)   # reference sampler
ENV["NEXTGP_HIP"] = "1"; runLMEM(f, data, nChain, nBurn, nThin; VCV = priors, outFolder = gpudir)   # libnextgp_hip
bref = vec(mean(readdlm(joinpath(refdir, "betaMOut"); skipstart = 1), dims = 1))
bgpu = vec(mean(readdlm(joinpath(gpudir, "betaMOut"); skipstart = 1), dims = 1))
eref = mean(readdlm(joinpath(refdir, "varEOut"); skipstart = 1))
egpu = mean(readdlm(joinpath(gpudir, "varEOut"); skipstart = 1))
println("cor(posterior mean beta)      = ", cor(bref, bgpu))
println("posterior mean varE ref / gpu = ", eref, " / ", egpu, "  (relative difference ", abs(eref - egpu) / eref, ")")
