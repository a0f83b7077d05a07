import sys, time, torch, os
sys.path.insert(0, os.getcwd())
import bench
print("cpus", os.cpu_count())
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    r = bench.cpu_baseline("chignolin", 600, 2, 2)
    print(th, r["value"], r["sample"][-60:], flush=True)
for th in (8, 16, 32):
    torch.set_num_threads(th)
    r = bench.cpu_baseline("dipeptide", 600, 32, 2)
    print("dip", th, r["value"], r["sample"][-60:], flush=True)
