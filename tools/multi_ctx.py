"""One GPU, G contexts on G host threads, the 75-frame track cut into G contiguous frame ranges (sac_amd.dist.frame_shard): do
the groups' optimizer rounds fill each other's bubbles? usage: multi_ctx.py [frames] G..."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from sac_amd import sacopt
from sac_amd.dist import frame_shard
from sac_amd.synth import stereo_clip
FRAME = 176400
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 75
L, R = stereo_clip(1000, NF * FRAME)
dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
o = [torch.empty_like(dL) for _ in range(6)]
frames = [(dL.data_ptr() + 4 * f * FRAME, dR.data_ptr() + 4 * f * FRAME, FRAME) for f in range(NF)]
outs = [{k: o[i].data_ptr() + 4 * f * FRAME for i, k in enumerate(("e0", "e1", "u0", "u1", "m0", "m1"))} for f in range(NF)]
ref = None
for G in [int(a) for a in sys.argv[2:]] or [1, 2, 3]:
    ctxs = [sacopt.Context(0) for _ in range(G)]
    best = [None] * G
    def work(g):
        lo, hi = frame_shard(NF, G, g)
        best[g], _ = ctxs[g].encode(2, frames[lo:hi], device_out=outs[lo:hi])
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(g,)) for g in range(G)]
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    b = np.concatenate(best)
    chk = (b.tobytes(), o[0].cpu().numpy().tobytes()[:1 << 20], o[5].cpu().numpy().tobytes()[-(1 << 20):])
    if ref is None: ref = chk
    print(f"G={G}: {dt * 1e3:8.1f} ms = {NF * FRAME / dt / 1e6:6.2f} M samples/s   same results as G=1: {chk == ref}", flush=True)
    for c in ctxs: c.close()
