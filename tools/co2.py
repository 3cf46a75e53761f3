"""Co-residency of the two-channel (1280,640) kernel: N chains x 4096 samples through the high cascade with the FIXTURE's
Intel rsqrtss table (10 bits: a build with -DRSQ_LDS_MAX=2048 -DS2_CHP=128 keeps the kernel's LDS under 80 KB, so two workgroups
share a CU). Prints the lead stage's summed launch time. usage: SACOPT_LIB=... co2.py N..."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from sac_amd import sacopt
from sac_amd.synth import stereo_clip
FR = 4096
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rsqrt_table.npz"))
ctx = sacopt.Context(0) if os.environ.get("CO2_HOST") else sacopt.Context(0, d["tab"].astype(np.float32), int(d["bits"]))  # CO2_HOST=1: the host's own table (12 bits on an EPYC: 16 KB of LDS)
ref = None
for N in [int(a) for a in sys.argv[1:]] or [224, 448]:
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
    chk = ob[0][:FR].cpu().numpy().tobytes()
    print(f"lib={os.path.basename(os.path.dirname(sacopt.LIB_PATH))} N={N:4d} chains x {FR}: call {dt*1e3:8.1f} ms, lead stage {lt['lms_ms']:8.2f} ms in "
          f"{lt['lms_launches']} launch(es) = {lt['lms_ms']*2.4e6/FR/N*256:8.0f} CU-cycles per stereo sample at 256 CUs  e0[:64] crc {hash(chk) & 0xffff:04x}", flush=True)
ctx.close()
