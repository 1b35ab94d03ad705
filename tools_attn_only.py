import sys, torch
sys.path.insert(0, ".")
from bench import measure_window_attention
print(measure_window_attention(16, 512, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 20))
