"""Independent numpy restatement of the reference's marker-effect Gibbs iteration (TEST INFRASTRUCTURE).

Written from the Julia source, not from oracle/ngp_oracle.c, so that a transcription error in the C oracle does not
pass unnoticed ("parity" item 1): data structures keep the reference's shape -- M[set] holds
data / Mp / mpm / lhs / rhs / regionArray / scale / df / logPi / piHat / estPi, beta and delta are per-set row vectors,
varBeta[set] a vector -- and every statement cites the line it restates:

    /root/reference/src/samplers.jl:29-53      iteration order (varE, fixed effects, marker sets)
    /root/reference/src/functions.jl:39-47     sampleX!, single column (the intercept)
    /root/reference/src/functions.jl:118-137   sampleBayesPR!(::Symbol)
    /root/reference/src/functions.jl:140-154   sampleBayesPR!(::Tuple), :513-516 sampleVarCovBetaPR, set-up src/mme.jl:448-489
    /root/reference/src/functions.jl:157-195   sampleBayesB!
    /root/reference/src/functions.jl:197-235   sampleBayesC!
    /root/reference/src/functions.jl:238-289   sampleBayesR!
    /root/reference/src/functions.jl:493-495, 509-511, 523-525, 531-533   sampleBeta, sampleVarBetaPR, sampleVarE, samplePi

The only thing shared with the oracle is the draw layer (the reference's own draws come from Distributions.jl / Random,
absent from /root/reference): `draw(kind, index, what, p1, p2)` returns the first draw of the keyed stream, exactly the
number the oracle and the device use at that site.  Dots are numpy's (pairwise summation), so results agree with the C
oracle to rounding, not bit for bit.
"""
import math

import numpy as np

KIND = dict(VARE_CHI2=1, FIXED_NORMAL=2, BETA_NORMAL=3, REGION_CHI2=4, B_UNIFORM=5, B_LOCUS_CHI2=6, PI_BETA=7, R_UNIFORM=8, R_DIRICHLET=9,
            T_WISHART=11)


class RefChain:
    def __init__(self, O, X, y, seed, chain, intercept=True):
        self.O, self.seed, self.chain = O, seed, chain
        self.N = len(y)
        self.X = np.asarray(X, dtype=np.float64)            # M[set].data is a Float64 matrix (prepMatVec.jl:116-129)
        self.ycorr = np.array(y, dtype=np.float64)          # mme.jl:57
        self.intercept = intercept
        self.ones = np.ones(self.N)                         # prepMatVec.jl:163-165
        self.b = np.zeros(1)
        self.M, self.beta, self.delta, self.varBeta = [], [], [], []
        self.E_df, self.E_scale = 4.0, 0.0005               # mme.jl:87-93
        self.iter = 0
        self.varE = float("nan")

    def draw(self, kind, index, what, p1=0.0, p2=0.0):
        return float(self.O.draws(self.seed, self.chain, self.iter, KIND[kind], index, what, 1, p1, p2, indexed=True)[0])

    def add_set(self, col0, ncol, method, df, scale, regions, varBeta0, pi0=0.0, estPi=False, lhs=None, rhs=None):
        data = self.X[:, col0:col0 + ncol]
        m = dict(method=method, data=data, Mp=[data[:, j].copy() for j in range(ncol)],          # mme.jl:308: a second copy
                 mpm=[float(np.dot(data[:, j], data[:, j])) for j in range(ncol)],               # mme.jl:305-307
                 lhs=np.zeros(ncol) if lhs is None else np.array(lhs, float),                     # mme.jl:314-322
                 rhs=np.zeros(ncol) if rhs is None else np.array(rhs, float),
                 regionArray=[range(a, b) for a, b in regions], scale=scale, df=df, estPi=estPi, dims=(self.N, ncol),
                 piHat=np.array([1.0 - pi0, pi0]), logPi=np.array([math.log(1.0 - pi0), math.log(pi0)]) if pi0 > 0 else np.zeros(2))
        self.M.append(m)
        self.beta.append(np.zeros(ncol))                     # mme.jl:443
        self.delta.append(np.ones(ncol, dtype=np.int64))     # mme.jl:444
        self.varBeta.append(np.array(varBeta0, dtype=np.float64))  # mme.jl:516

    # ---- functions.jl ----
    def sampleVarE(self):                                    # :523-525
        return (self.E_df * self.E_scale + np.dot(self.ycorr, self.ycorr)) / self.draw("VARE_CHI2", 0, 2, self.E_df + self.N)

    def sampleX(self, varE):                                 # :39-47, length(b) == 1
        iVarE = 1.0 / varE
        self.ycorr += self.ones * self.b[0]
        rhs = np.dot(self.ones, self.ycorr) * iVarE + 0.0
        lhs = float(self.N) * iVarE + 0.0                    # xpx = N (mme.jl:138)
        meanMu = rhs / lhs
        self.b[0] = meanMu + math.sqrt(1.0 / lhs) * self.draw("FIXED_NORMAL", 0, 1)
        self.ycorr -= self.ones * self.b[0]

    def add_fixed(self, X, lhs=None, rhs=None):              # mme.jl:120-152
        X = np.asarray(X, float)
        X = X[:, None] if X.ndim == 1 else X
        nCol = X.shape[1]
        xpx = X.T @ X                                                                             # :137
        if nCol > 1:
            xpx = xpx + np.eye(nCol) * (np.abs(np.diag(xpx)) / 10000).min()                       # :149-152
        self.Xfix = getattr(self, "Xfix", [])
        self.Xfix.append(dict(data=X, Xp=X.T.copy(), xpx=xpx, nCol=nCol, b=np.zeros(nCol),
                              lhs=np.zeros(nCol) if lhs is None else np.array(lhs, float), rhs=np.zeros(nCol) if rhs is None else np.array(rhs, float)))

    def sampleXset(self, f, varE):                           # :39-53
        Xs = self.Xfix[f]
        iVarE = 1.0 / varE
        b = Xs["b"]
        if Xs["nCol"] == 1:
            self.ycorr += Xs["data"][:, 0] * b[0]
            rhs = float(Xs["Xp"][0] @ self.ycorr) * iVarE + Xs["rhs"][0]
            lhs = float(Xs["xpx"][0, 0]) * iVarE + Xs["lhs"][0]
            b[0] = rhs / lhs + math.sqrt(1.0 / lhs) * self.draw("FIXED_NORMAL", (f + 1) << 20, 1)
            self.ycorr -= Xs["data"][:, 0] * b[0]
        else:
            self.ycorr += Xs["data"] @ b                                                          # :49
            bVec = b.copy()                                                                       # :22-36
            Yi = Xs["Xp"] @ self.ycorr * iVarE
            for i in range(Xs["nCol"]):
                bVec[i] = 0.0
                rhsb = Yi[i] - np.dot(Xs["xpx"][i, :], bVec) * iVarE
                lhsb = Xs["xpx"][i, i] * iVarE
                invLhsb = 1.0 / lhsb
                bVec[i] = invLhsb * rhsb + math.sqrt(invLhsb) * self.draw("FIXED_NORMAL", ((f + 1) << 20) | i, 1)
            b[:] = bVec
            self.ycorr -= Xs["data"] @ b                                                          # :51

    def sampleBeta(self, si, locus, meanBeta, lhs):          # :493-495
        return meanBeta + math.sqrt(1.0 / lhs) * self.draw("BETA_NORMAL", (si << 40) | locus, 1)

    def sampleBayesPR(self, si, varE):                       # :118-137
        M, beta, vb = self.M[si], self.beta[si], self.varBeta[si]
        iVarE = 1.0 / varE
        for r, theseLoci in enumerate(M["regionArray"]):
            regionSize = len(theseLoci)
            iVarBeta = 1.0 / vb[r]
            for locus in theseLoci:
                self.ycorr += beta[locus] * M["data"][:, locus]                                   # :128
                rhs = np.dot(M["Mp"][locus], self.ycorr) * iVarE + M["rhs"][locus]                # :129
                lhs = M["mpm"][locus] * iVarE + M["lhs"][locus] + iVarBeta                        # :130
                meanBeta = rhs / lhs                                                              # :131
                beta[locus] = self.sampleBeta(si, locus, meanBeta, lhs)                           # :132
                self.ycorr += -1.0 * beta[locus] * M["data"][:, locus]                            # :133
            sub = beta[theseLoci.start:theseLoci.stop]
            vb[r] = (M["scale"] * M["df"] + np.dot(sub, sub)) / self.draw("REGION_CHI2", (si << 40) | r, 2, M["df"] + regionSize)  # :135

    def _inclusion(self, si, locus, rrr, varE, vbeta):       # :169-174 / :210-217
        M = self.M[si]
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            v0 = np.float64(M["mpm"][locus]) * varE
            v1 = (np.float64(M["mpm"][locus]) ** 2) * vbeta + v0
            logDelta0 = -0.5 * (np.log(v0) + (rrr ** 2) / v0) + M["logPi"][0]
            logDelta1 = -0.5 * (np.log(v1) + (rrr ** 2) / v1) + M["logPi"][1]
            probDelta1 = 1.0 / (1.0 + np.exp(logDelta0 - logDelta1))
        return self.draw("B_UNIFORM", (si << 40) | locus, 0) < probDelta1                         # NaN compares false

    def samplePi(self, si, nLoci):                           # :531-533, then :190-194
        M = self.M[si]
        piIn = self.draw("PI_BETA", si, 3, nLoci + 1.0, M["dims"][1] - nLoci + 1.0)
        M["piHat"][:] = [1.0 - piIn, piIn]
        M["logPi"][:] = np.log(M["piHat"])

    def sampleBayesB(self, si, varE):                        # :157-195
        M, beta, delta, vb = self.M[si], self.beta[si], self.delta[si], self.varBeta[si]
        nLoci = 0
        for r, theseLoci in enumerate(M["regionArray"]):     # one region per locus: r == locus
            iVarE = 1.0 / varE
            with np.errstate(divide="ignore"):
                iVarBeta = np.float64(1.0) / vb[r]            # :165, Inf after an exclusion
            for locus in theseLoci:
                self.ycorr += beta[locus] * M["data"][:, locus]                                   # :167
                rrr = np.dot(M["data"][:, locus], self.ycorr)                                     # :168
                if self._inclusion(si, locus, rrr, varE, vb[r]):                                  # :174
                    delta[locus] = 1
                    nLoci += 1
                    rhs = np.dot(M["Mp"][locus], self.ycorr) * iVarE + M["rhs"][locus]            # :177
                    lhs = M["mpm"][locus] * iVarE + M["lhs"][locus] + iVarBeta                    # :178
                    meanBeta = rhs / lhs
                    beta[locus] = self.sampleBeta(si, locus, meanBeta, lhs)                       # :180
                    self.ycorr += -1.0 * beta[locus] * M["data"][:, locus]                        # :181
                    vb[r] = (M["scale"] * M["df"] + beta[locus] * beta[locus]) / self.draw("B_LOCUS_CHI2", (si << 40) | locus, 2, M["df"] + 1.0)  # :182
                else:
                    beta[locus] = 0.0                                                             # :184
                    delta[locus] = 0
                    vb[r] = 0.0                                                                   # :186
        if M["estPi"]:
            self.samplePi(si, nLoci)

    def sampleBayesC(self, si, varE):                        # :197-235
        M, beta, delta, vb = self.M[si], self.beta[si], self.delta[si], self.varBeta[si]
        nLoci = 0
        iVarE = 1.0 / varE
        iVarBeta = 1.0 / vb[0]                                                                    # :205
        for locus in range(M["dims"][1]):
            self.ycorr += beta[locus] * M["data"][:, locus]                                       # :208
            rrr = np.dot(M["data"][:, locus], self.ycorr)                                         # :209
            if self._inclusion(si, locus, rrr, varE, vb[0]):                                      # :217
                delta[locus] = 1
                nLoci += 1
                rhs = np.dot(M["Mp"][locus], self.ycorr) * iVarE                                  # :220 (M.rhs commented out)
                lhs = M["mpm"][locus] * iVarE + M["lhs"][locus] + iVarBeta                        # :221
                meanBeta = rhs / lhs
                beta[locus] = self.sampleBeta(si, locus, meanBeta, lhs)
                self.ycorr += -1.0 * beta[locus] * M["data"][:, locus]                            # :224
            else:
                beta[locus] = 0.0
                delta[locus] = 0
        vb[0] = (M["scale"] * M["df"] + np.dot(beta, beta)) / self.draw("REGION_CHI2", (si << 40), 2, M["df"] + nLoci)  # :231
        if M["estPi"]:
            self.samplePi(si, nLoci)

    def add_set_r(self, col0, ncol, df, scale, varBeta0, vClass, pi, estPi=False, lhs=None, rhs=None):   # mme.jl:374-383
        self.add_set(col0, ncol, 3, df, scale, [(j, j + 1) for j in range(ncol)], [varBeta0], pi0=0.5, estPi=estPi, lhs=lhs, rhs=rhs)
        M = self.M[-1]
        M["vClass"] = np.array(vClass, float); M["piHat"] = np.array(pi, float); M["logPi"] = np.log(M["piHat"])

    def sampleBayesR(self, si, varE):                        # :238-289
        M, beta, delta, vb = self.M[si], self.beta[si], self.delta[si], self.varBeta[si]
        nVarClass = len(M["vClass"])
        nLoci = np.zeros(nVarClass, dtype=np.int64)
        nNonZero = 0
        varc = vb[0] * M["vClass"]                                                                # :244
        sumS = 0.0
        iVarE = 1.0 / varE
        for r, theseLoci in enumerate(M["regionArray"]):
            for locus in theseLoci:
                self.ycorr += beta[locus] * M["data"][:, locus]                                   # :249
                rhs = np.dot(M["Mp"][locus], self.ycorr) * iVarE + M["rhs"][locus]                # :250
                lhs = np.zeros(nVarClass); ExpLogL = np.zeros(nVarClass)
                for v in range(nVarClass):
                    lhs[v] = 0.0 if varc[v] == 0.0 else M["mpm"][locus] * iVarE + M["lhs"][locus] + 1.0 / varc[v]                 # :254
                    logLc = M["logPi"][v] if varc[v] == 0.0 else -0.5 * (math.log(varc[v] * lhs[v]) - ((rhs ** 2) / lhs[v])) + M["logPi"][v]  # :255
                    ExpLogL[v] = math.exp(logLc)
                probs = ExpLogL / ExpLogL.sum()                                                   # :259
                cumProbs = np.cumsum(probs)
                classSNP = nVarClass - 1
                for v in range(nVarClass):                    # findfirst(x -> x >= rand(), cumProbs): a fresh uniform per comparison
                    if cumProbs[v] >= self.draw("R_UNIFORM", (si << 40) | (locus << 3) | v, 0):
                        classSNP = v
                        break
                delta[locus] = classSNP + 1                                                       # :262
                nLoci[classSNP] += 1
                if varc[classSNP] != 0.0:                                                         # :265
                    nNonZero += 1
                    meanBeta = rhs / lhs[classSNP]
                    betaSample = self.sampleBeta(si, locus, meanBeta, lhs[classSNP])
                    beta[locus] = betaSample
                    self.ycorr += -1.0 * beta[locus] * M["data"][:, locus]                        # :270
                    sumS += betaSample ** 2 / M["vClass"][classSNP]                               # :272-273
                else:
                    beta[locus] = 0.0
        vb[0] = (M["scale"] * M["df"] + sumS) / self.draw("REGION_CHI2", (si << 40), 2, M["df"] + nNonZero)   # :281, :518-520
        if M["estPi"]:                                                                            # :284-288, :536-538
            g = np.array([self.O.draws(self.seed, self.chain, self.iter, KIND["R_DIRICHLET"], (si << 40) | v, 4, 1, nLoci[v] + 1.0, indexed=True)[0]
                          for v in range(nVarClass)])
            M["piHat"][:] = g / g.sum()
            M["logPi"][:] = np.log(M["piHat"])

    def add_set_tuple(self, cols, df, scale, regions, v):    # mme.jl:448-489; cols[l, m] = panel column of component m of locus l
        cols = np.asarray(cols)
        nloc, k = cols.shape
        data = [self.X[:, cols[l, :]] for l in range(nloc)]                                       # :456-457: X_l, N x k
        m = dict(method=4, k=k, cols=cols, data=data, Mp=[d.T.copy() for d in data],             # :463
                 mpm=[d.T @ d for d in data],                                                     # :462
                 regionArray=[range(a, b) for a, b in regions], scale=np.array(scale, float).reshape(k, k), df=df,
                 piHat=np.array([0.5, 0.5]), span=int(cols.max() - cols.min() + 1), col0=int(cols.min()))
        self.M.append(m)
        self.beta.append(np.zeros((k, nloc)))                 # one row vector per set of the tuple (beta[M.pos])
        self.delta.append(np.ones(m["span"], dtype=np.int64))
        self.varBeta.append([np.array(v, float).reshape(k, k).copy() for _ in regions])           # :516

    def sampleBayesPR_tuple(self, si, varE):                  # functions.jl:140-154
        M, beta, vb = self.M[si], self.beta[si], self.varBeta[si]
        k = M["k"]
        for r, theseLoci in enumerate(M["regionArray"]):
            regionSize = len(theseLoci)
            invB = np.linalg.inv(vb[r])                                                           # :143
            for locus in theseLoci:
                self.ycorr += M["data"][locus] @ beta[:, locus]                                   # :145
                RHS = (M["Mp"][locus] @ self.ycorr) / varE                                        # :146
                invLHS = np.linalg.inv(M["mpm"][locus] / varE + invB)                             # :147
                meanBETA = invLHS @ RHS                                                           # :148
                z = np.array([self.draw("BETA_NORMAL", (si << 40) | (locus * k + m), 1) for m in range(k)])
                L = np.linalg.cholesky((invLHS + invLHS.T) / 2)                                   # rand(MvNormal(mean, Symmetric(invLHS))) = mean + L z
                beta[:, locus] = meanBETA + L @ z                                                 # :149
                self.ycorr -= M["data"][locus] @ beta[:, locus]                                   # :150
            B = beta[:, theseLoci.start:theseLoci.stop].T                                         # reduce(hcat, ...): regionSize x k
            Sb = B.T @ B                                                                          # :514
            vb[r] = self.inverse_wishart(si, r, M["df"] + regionSize, M["scale"] + Sb)            # :515

    def inverse_wishart(self, si, r, nu, Psi):
        """rand(InverseWishart(nu, Psi)) by Bartlett's construction on the keyed draws (Distributions.jl is absent): W = (L A)(L A)',
        L = chol(inv(Psi)), A lower, A_ii = sqrt(chi2(nu - i)), A_ij ~ N(0, 1); result inv(W).  k = 1: Psi / chi2(nu)."""
        k = Psi.shape[0]
        if k == 1:
            return Psi / self.draw("REGION_CHI2", (si << 40) | r, 2, nu)
        L = np.linalg.cholesky(np.linalg.inv(Psi))
        A = np.zeros((k, k))
        for i in range(k):
            for j in range(i + 1):
                key = ("REGION_CHI2", (si << 40) | r) if i == 0 else ("T_WISHART", (si << 40) | (r << 8) | (i << 4) | j)
                A[i, j] = math.sqrt(self.draw(key[0], key[1], 2, nu - i)) if i == j else self.draw(key[0], key[1], 1)
        LA = L @ A
        return np.linalg.inv(LA @ LA.T)

    # ---- samplers.jl:29-53 ----
    def run(self, niter):
        for _ in range(niter):
            self.iter += 1
            varE = self.sampleVarE()                         # :32-35
            self.varE = varE
            if self.intercept:
                self.sampleX(varE)                           # :39-41
            for f in range(len(getattr(self, "Xfix", []))):
                self.sampleXset(f, varE)
            for si, M in enumerate(self.M):                  # :50-53
                {0: self.sampleBayesPR, 1: self.sampleBayesB, 2: self.sampleBayesC, 3: self.sampleBayesR, 4: self.sampleBayesPR_tuple}[M["method"]](si, varE)

    def _panel_beta(self, si):
        M = self.M[si]
        if M["method"] != 4:
            return self.beta[si]
        out = np.zeros(M["span"])                              # the tuple's effects at their panel columns (unused columns 0)
        out[M["cols"] - M["col0"]] = self.beta[si].T
        return out

    def state(self):
        return dict(ycorr=self.ycorr.copy(), beta=np.concatenate([self._panel_beta(si) for si in range(len(self.M))]), delta=np.concatenate(self.delta),
                    varBeta=np.concatenate([np.concatenate([m.ravel() for m in vb]) if self.M[si]["method"] == 4 else vb for si, vb in enumerate(self.varBeta)]),
                    piHat=np.concatenate([m["piHat"] if m["method"] != 3 else np.array([0.5, 0.5]) for m in self.M]),
                    class_pi=[m["piHat"].copy() if m["method"] == 3 else None for m in self.M], varE=self.varE, b=float(self.b[0]))
