"""The launches tools/hbm_traffic.sh counts fabric traffic on: window attention forward / backward and the MLP's 17-tap convolution at the
benchmark geometry (a handful of each: the counter pass replays every kernel).  python tools/traffic_cmd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

bench.measure_window_attention(16, 512, iters=4)
bench.measure_mlp_conv(16, 512, iters=4)
bench.measure_dominant_kernels(16, 512, iters=3)
