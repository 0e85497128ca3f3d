import sys, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ngp_pkg import load_pkg
ngp = load_pkg()
lag = int(sys.argv[1]) if len(sys.argv)>1 else 6
N = int(sys.argv[2]) if len(sys.argv)>2 else 10000
P = int(sys.argv[3]) if len(sys.argv)>3 else 100000
s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=lag, streamer=int(sys.argv[4]) if len(sys.argv) > 4 else None,
              storage=os.environ.get("NGP_TOOL_STORAGE"))
if "NGP_TOOL_CHAIN_FORM" in os.environ: s.set_chain_form(int(os.environ["NGP_TOOL_CHAIN_FORM"]))
if "NGP_TOOL_KNOB" in os.environ: s.debug_set_knob(int(os.environ["NGP_TOOL_KNOB"]))
if "NGP_TOOL_NEAR" in os.environ: s.set_near(int(os.environ["NGP_TOOL_NEAR"]))
s.generate_panel(N,P)
rng=np.random.default_rng(1); bt=np.zeros(P); idx=rng.choice(P,P//100,replace=False); bt[idx]=rng.normal(size=P//100)
g=s.xbeta(bt); y=10+g+np.random.default_rng(2).normal(size=N)*np.sqrt(g.var())
v=0.5*y.var()/(s.mpm().sum()/N)
tk = int(os.environ.get("NGP_TOOL_TUPLE", "0"))  # k: the panel as ONE correlated (Tuple BayesPR) set of k components per locus
if os.environ.get("NGP_TOOL_METHOD") == "R":
    s.add_marker_set_r(0, P, 4.0, v * 0.5, v, [0.0, 0.01, 0.1, 1.0], [0.95, 0.03, 0.015, 0.005], estPi=True)
elif tk:
    nloc = (P // 64) * (64 // tk); V = v * (0.7 * np.eye(tk) + 0.3)
    s.add_marker_set_tuple(0, nloc, tk, 3.0 + tk, V * 0.5, [(0, nloc)], V)
else:
    s.add_marker_set(0,P,0,4.0,v*0.5,[(0,P)],[v])
s.set_y(y); s.set_residual_prior(4.0,0.25*y.var())
s.run(3)
t=time.perf_counter(); s.run(10); dt=(time.perf_counter()-t)/10
print(f"lag {lag}: {dt*1e3:.2f} ms/iter  layout {s.layout()}")
s.debug_stamps(True); s.run(1)
d = s.debug_stamps(True, n=(1<<20)+2*(s.layout()[2]+16)).astype(np.int64)
nb = s.layout()[2]; lag = s.config()[1]
u_lo, u_hi = nb//8, nb - nb//8
S = d[:4*nb].reshape(nb,4)
T = d[(1<<20):(1<<20)+2*(nb+lag)].reshape(nb+lag,2)
u = np.arange(u_lo,u_hi)
start, chain_end, pub = S[u,0], S[u,1], S[u,2]
print("per-block period (start[u+1]-start[u]) us:", np.median(np.diff(S[u_lo:u_hi,0]))/100.0)
print("chain time us:", np.median(chain_end-start)/100.0)
print("publish after chain end us:", np.median(pub-chain_end)/100.0)
# streamer 0: time partial(u) published vs sampler block start
print("streamer0 partial(u) published before sampler start(u) by us:", np.median(S[u,0]-T[u,0])/100.0)
print("streamer0 got dlt(u-D) after publish(u-D) us:", np.median(T[u+lag,1]-S[u,2])/100.0)
print("streamer0 loop period us:", np.median(np.diff(T[u_lo:u_hi,0]))/100.0)

W = d[(1<<19):(1<<19)+8*nb].reshape(nb,8)
for w in range(8):
    ue, uo = u[u % 2 == 0], u[u % 2 == 1]
    print(f"wave {w} end-of-work after block start us:", np.median(W[u,w]-S[u,0])/100.0,
          " even blocks", np.median(W[ue,w]-S[ue,0])/100.0, " odd blocks", np.median(W[uo,w]-S[uo,0])/100.0)
print("slowest wave per block (histogram):", np.bincount(np.argmax(W[u], axis=1), minlength=8))
W5 = d[(5<<17):(5<<17)+4*nb].reshape(nb,4)
print("lag-1 wave: its Gram rows arrived after block start us:", np.median(W5[u,0]-S[u,0])/100.0, " 10/90 %:", np.percentile(W5[u,0]-S[u,0],10)/100.0, np.percentile(W5[u,0]-S[u,0],90)/100.0)
print("lag-1 wave: product done after block start us:", np.median(W5[u,1]-S[u,0])/100.0, " (rows -> product:", np.median(W5[u,1]-W5[u,0])/100.0, ")")
print("lag-1 wave: rows asked for (end of work of the previous block) -> arrived us:", np.median(W5[u,0]-W[u-1,5])/100.0)
print("chain wave: has its total after block start us:", np.median(W5[u,2]-S[u,0])/100.0, " (lag-1 product done -> total read:", np.median(W5[u,2]-W5[u,1])/100.0, ")")
print("chain wave: diagonal block in registers after its total us:", np.median(W5[u,3]-W5[u,2])/100.0, "; the 64 steps + publication (block in registers -> chain end) us:", np.median(S[u,1]-W5[u,3])/100.0)

Rd = d[(3<<18):(3<<18)+2*nb].reshape(nb,2)
print("reducer0: counter complete for block u, relative to sampler start(u) us:", np.median(Rd[u,0]-S[u,0])/100.0)
print("reducer0: publish latency (complete -> gsum published) us:", np.median(Rd[u,1]-Rd[u,0])/100.0)
print("streamer0 publish(u) -> reducer0 complete(u) us:", np.median(Rd[u,0]-T[u,0])/100.0)
print("wave2 poll success for block u relative to sampler start(u) us:", np.median(S[u,3]-S[u,0])/100.0)
print("reducer0 published(u) -> wave2 poll success(u) us:", np.median(S[u,3]-Rd[u,1])/100.0)

Sn = s.layout()[1]
Aall = d[(7<<17):(7<<17)+4*Sn].reshape(Sn,4)
print("sampler xcc", d[(7<<17)-2], "hwid", hex(int(d[(7<<17)-1])))
for col,blk in ((0,800),(2,1200)) if nb > 1300 else ():
    rel = (Aall[:,col]-S[blk,0])/100.0
    order = np.argsort(rel)[-4:]
    print("block",blk,"publish rel sampler start: min %.1f med %.1f max %.1f" % (rel.min(), np.median(rel), rel.max()), "slowest:", [(int(i), round(float(rel[i]),1), int(Aall[i,1])-1, hex(int(Aall[i,3]))) for i in order])

Acc = d[(7<<17)+4096:(7<<17)+4096+8*Sn].reshape(Sn,8)[:, :5].astype(np.float64)/100.0/(nb+lag)
tot = Acc[:,0]+Acc[:,1]+Acc[:,2]
print("per-block avg us  phaseA(wait) phaseB(update) phaseC(gemv+pub)  w7poll  w4dma :  median", np.round(np.median(Acc,axis=0),2))
worst = np.argsort(Acc[:,3])[-3:]
for i in worst: print("   streamer", int(i), "xcc", int(Aall[i,1])-1, np.round(Acc[i],2))
best = np.argsort(Acc[:,3])[:2]
for i in best: print("   streamer", int(i), "xcc", int(Aall[i,1])-1, np.round(Acc[i],2))

# raw timeline of a few consecutive blocks (us relative to the block's start stamp): where the period goes
print("raw: block | next start | chain end | dlt published | sum in r0 (this block's) | end-of-work of waves 0..7")
for ub in range(nb // 2, nb // 2 + 10):
    s0 = S[ub, 0]
    print(f"  {ub:6d} | {(S[ub + 1, 0] - s0) / 100.0:6.2f} | {(S[ub, 1] - s0) / 100.0:6.2f} | {(S[ub, 2] - s0) / 100.0:6.2f} | {(S[ub, 3] - s0) / 100.0:7.2f} | "
          + " ".join(f"{(W[ub, w] - s0) / 100.0:6.2f}" for w in range(8)))
