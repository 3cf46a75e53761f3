"""Throughput of the (1280,640) stage kernels when workgroups share CUs: N short frames through the
whole cascade (SACOPT_DBG_ALL=512: one workgroup per channel, 256: one per chain)."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ablate", "libsacopt.so"))  # instrumented build: make -C tools
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FR = 4096
ctx = sacopt.Context(0)
for N in [int(a) for a in sys.argv[1:]] or [112, 224, 448]:
    L, R = stereo_clip(1000, N * FR)
    dL = torch.from_numpy(L).cuda(); dR = torch.from_numpy(R).cuda()
    frames = [(dL.data_ptr() + 4 * f * FR, dR.data_ptr() + 4 * f * FR, FR) for f in range(N)]
    ob = [torch.empty_like(dL) for _ in range(4)]
    outs = [dict(e0=ob[0].data_ptr() + 4 * f * FR, e1=ob[1].data_ptr() + 4 * f * FR,
                 u0=ob[2].data_ptr() + 4 * f * FR, u1=ob[3].data_ptr() + 4 * f * FR) for f in range(N)]
    ctx.predict(2, frames, device_out=outs); torch.cuda.synchronize()
    t = time.perf_counter(); ctx.predict(2, frames, device_out=outs); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"N={N:4d} chains x {FR}: {dt*1e3:8.1f} ms  ({dt*2.4e9/FR:8.0f} cycles per sample-step of the whole launch)", flush=True)

ctx.close()
