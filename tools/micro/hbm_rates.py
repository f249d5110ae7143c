"""Plain streaming rates of the box's HBM as the runtime's own fill / copy kernels see them (GPU box):
python tools/micro/hbm_rates.py  -- fill (write only), copy (read + write) and sum (read only) of a 34.4 GB buffer: what a
kernel that only writes, reads and writes, or only reads a C4 field can hope for at best."""
import time, torch
n = 2048 * 2048 * 1025 * 2          # floats of one half-spectrum field
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
gb = n * 4 / 1e9
t = timed(lambda: a.zero_());            print("fill  %.1f GB: %.2f ms = %.2f TB/s written" % (gb, t, gb / t))
t = timed(lambda: b.copy_(a));           print("copy  %.1f GB: %.2f ms = %.2f TB/s read + written" % (gb, t, 2 * gb / t))
t = timed(lambda: a.sum());              print("sum   %.1f GB: %.2f ms = %.2f TB/s read" % (gb, t, gb / t))
