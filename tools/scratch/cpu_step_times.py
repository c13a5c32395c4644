import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from oracle import model_ref
print("cpus", os.cpu_count(), open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0])
for threads in (int(sys.argv[1]) if len(sys.argv) > 1 else 64,):
    torch.set_num_threads(threads)
    t = time.time()
    r = model_ref.timed_training_sample(os.path.join(ROOT, bench.YAML), 100, 1024, 2048, 2, bench.benchmark_init, budget_s=1e9)
    print(threads, json.dumps(r), "total", time.time() - t)
