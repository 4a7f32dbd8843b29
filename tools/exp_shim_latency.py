import importlib, sys, time, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
h, d = synth.synth_frame(100007, "crop")
s = {"header": h, "data": d}
for _ in range(10): r = pkg.cal_tsdf_cuda(s)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); pkg.cal_tsdf_cuda(s); ts.append(time.perf_counter() - t0)
print(f"cal_tsdf_cuda crop {h[4]-h[2]}x{h[5]-h[3]}: median {np.median(ts)*1e6:.1f} us, min {min(ts)*1e6:.1f} us")
h, d = synth.synth_frame(3, "full")
s = {"header": h, "depth": d}
for _ in range(10): r = pkg.cal_tsdf_cuda(s)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); pkg.cal_tsdf_cuda(s); ts.append(time.perf_counter() - t0)
print(f"cal_tsdf_cuda full frame: median {np.median(ts)*1e6:.1f} us")
h, d = synth.synth_frame(100007, "crop")
pc2 = np.array([[-60.0, -70.0, -480.0], [70.0, 60.0, -380.0]], np.float32)
for _ in range(10): pkg.tsdf_f({"header": h, "depth": d}, pc2)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); pkg.tsdf_f({"header": h, "depth": d}, pc2); ts.append(time.perf_counter() - t0)
print(f"tsdf_f (loop entry, float64 [c,x,y,z] result): median {np.median(ts)*1e6:.1f} us")
