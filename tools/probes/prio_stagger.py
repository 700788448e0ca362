"""ECG B=512 ms/step for ffn_prio x ffn_stagger (fair, opposed-phase priorities make the two workgroups of a CU run in
lockstep; a start stagger of about one epilogue should then keep their epilogues apart)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from fastfourierdiffusion_amd import _native as N
dev = torch.device("cuda", 0)
model, sch, _ = bench.build_model(dev, "ecg")
ctx = model._ctx(); lib = ctx.lib
B, L, Cn = 512, model.max_len, model.n_channels
sch.set_timesteps(1000)
ts_c = (C.c_float * 1000)(*sch.timesteps.tolist())
x = torch.randn(B, L, Cn, device=dev)
s = N.current_stream_ptr(dev)
def run():
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, 10, 1, 0, None, 0, 0, s), ctx.handle, "w")
    torch.cuda.synchronize(); t0 = time.time()
    N.check(lib.ffd_sample_batch(ctx.handle, x.data_ptr(), B, ts_c, 1000, float(sch.step_size), 0, 60, 1, 0, None, 0, 0, s), ctx.handle, "r")
    torch.cuda.synchronize()
    return (time.time() - t0) / 60 * 1e3
for rep in range(2):
    for prio in (0, 3):
        row = []
        for st in (0, 300, 600, 940, 1400, 2000, 2800):
            lib.ffd_tune(b"ffn_prio", prio); lib.ffd_tune(b"ffn_stagger", st)
            row.append(f"{st}:{run():.3f}")
        print(f"prio={prio}: " + "  ".join(row), flush=True)
