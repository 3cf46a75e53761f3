"""Throughput of the (1280,640) stage kernels when workgroups share CUs: N short frames through the whole cascade
(instrumented build; SACOPT_DBG_ALL=512: k_lms1c, one workgroup per channel = 2N workgroups; 256: the two-channel kernel,
N workgroups). Prints the lead stage's launch time (HIP events on the library's stream) and the whole call.
SACOPT_LIB=tools/_nt512/libsacopt.so: k_lms1c with 512 threads (two workgroups fit a CU)."""
import os as _os; _os.environ.setdefault("SACOPT_LIB", _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "_ablate", "libsacopt.so"))  # instrumented build: make -C tools
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FR = 4096
torch.cuda.init()
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
    lt = ctx.last_timing()
    print(f"lib={os.path.basename(os.path.dirname(os.environ['SACOPT_LIB']))} dbg={os.environ.get('SACOPT_DBG_ALL','-')} N={N:4d} chains x {FR}: "
          f"call {dt*1e3:8.1f} ms, lead stage {lt['lms_ms']:8.2f} ms in {lt['lms_launches']} launch(es) = "
          f"{lt['lms_ms']*2.4e6/FR:8.0f} cycles per sample-step of the launch, {lt['lms_ms']*2.4e6/FR/N*256:8.0f} CU-cycles per stereo sample at 256 CUs", flush=True)
ctx.close()
