"""tools/hbm_traffic.sh table -> profiles JSON that bench.py's roofline.traffic reads, stamped with the hash of the kernel sources
the counters were taken on (bench.py reports null when the sources have changed since):
    python tools/hbm_traffic_json.py gpurun_out/<tag>.txt profiles/r02_hbm_traffic.json"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

src, dst = sys.argv[1], sys.argv[2]
out = {"_source": "%s (tools/hbm_traffic.sh: rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum in a "
                  "pass of its own; reads = RDREQ x 64 B x 2 per MI355X_MICROARCH.md (HBM, gfx950), writes = 32 B / 64 B requests; "
                  "B=16, 128x128 tokens, C=32, bf16)" % os.path.basename(src),
       "source_sha16": bench._source_sha16(*bench.ATTN_SOURCES), "mlp_source_sha16": bench._source_sha16(*bench.MLP_SOURCES)}
# key in the JSON -> text that identifies the table line (first match wins; the table is sorted by read requests)
NAMES = {"winattn_fwd_kernel": "winattn_fwd_kernel", "winattn_bwd_kernel": "winattn_bwd_kernel", "domega_reduce_kernel": "domega_reduce_kernel",
         "conv_taps128_kernel": "conv_taps128_kernel<1, false, false>",             # the MLP sum's forward launch (roofline_mfma.traffic)
         "conv_taps128_kernel_dgrad": "conv_taps128_kernel<1, true, false>",       # (in the step: with the BatchNorm-backward statistics)
         "conv_wgrad_planes_kernel": "conv_wgrad_planes_kernel", "conv_wgrad_pw_kernel_fused": "conv_wgrad_pw_kernel<4, 2, 2, 64, true, false",
         "bn_finapply_planes_kernel": "bn_finapply_planes_kernel"}
for line in open(src):
    for key, name in NAMES.items():
        if name in line and key not in out:
            f = line.split()
            out[key] = {"read_bytes": int(float(f[-2]) * 1e6), "write_bytes": int(float(f[-1]) * 1e6), "launches": int(f[-5])}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
