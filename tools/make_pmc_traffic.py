#!/usr/bin/env python3
"""profiles/r04_pmc_traffic.json from the table tools/pmc_traffic.py wrote (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
over `bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-configs --no-feed --no-p16`, tools/collect_profiles.sh): maps bench.py's
kernel groups to the kernel symbols that serve them and stamps the file with the source hash of the build it was
measured on (bench.py quotes `roofline.traffic` only when that hash matches the sources it runs).
   python tools/make_pmc_traffic.py gpurun_out/prof_<tag>/pmc_traffic_all.json [--batch 256 --filters 64]"""
import argparse, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

# group -> regex over "<kernel> grid=<n>" (one symbol per group in the headline model; the two data-gradient epilogue
# flavours of a resolution are one group)
GROUPS = {
    "conv3x3_wgrad@60x60": r"k_wgrad3x3_ps<64, true, false>",
    "conv3x3_wgrad@30x30": r"k_wgrad3x3_ps<32, true, false>",
    "conv3x3_wgrad@15x15": r"k_wgrad3x3_ps<16, true, false>",
    "conv3x3_fwd@60x60": r"k_conv3x3_ps<0, 64, false>",
    "conv3x3_fwd_pool@60x60": r"k_conv3x3_ps<2, 64, false>",
    "conv3x3_dgrad@60x60": r"k_conv3x3_ps<1, 64, false>",
    "conv3x3_dgrad_unpool@60x60": r"k_conv3x3_ps<3, 64, false>",
    "conv3x3_fwd@30x30": r"k_conv3x3_ps<0, 32, false>",
    "conv3x3_fwd_pool@30x30": r"k_conv3x3_ps<2, 32, false>",
    "conv3x3_dgrad@30x30": r"k_conv3x3_ps<1, 32, false>",
    "conv3x3_dgrad_unpool@30x30": r"k_conv3x3_ps<3, 32, false>",
    "chain_fwd@15x15": r"k_block_chain_ps<false, false>",
    "chain_bwd@15x15": r"k_block_chain_ps<true, false>",
    "stem_fwd@60x60": r"k_stem_fwd_x3_pipe",
    "stem_wgrad@60x60": r"k_stem_wgrad_x3_pipe",
    "pool_route_bwd@60x60": r"k_pool_route_bwd_ps.* grid=1843200",
    "pool_route_bwd@30x30": r"k_pool_route_bwd_ps.* grid=460800",
    "head_loss_fused@15x15": r"^k_head_fused grid",
}

ap = argparse.ArgumentParser()
ap.add_argument("table")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--filters", type=int, default=64)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_pmc_traffic.json"))
args = ap.parse_args()
tab = json.load(open(args.table))
kern = {}
for g, rx in GROUPS.items():
    hits = {k: v for k, v in tab.items() if re.search(rx, k)}
    if not hits:
        continue
    n = sum(v["launches"] for v in hits.values())
    kern[g] = {"kernel": " + ".join(hits), "launches": n,
               "hbm_bytes_per_launch": int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hits.values()) / n)}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over `bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-inference --no-configs --no-feed --no-p16`; HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies the "
               "128-B requests of 16-B-per-lane reads at 64 B, MI355X_MICROARCH.md HBM section; calibrated in round 2 on "
               "k_pool_route_bwd and k_stem_fwd_x3_pipe, whose byte counts are known: see DESIGN.md)",
       "csrc_sha256": bench.kernel_source_hash(), "batch": args.batch, "filters": args.filters, "source_table": os.path.basename(os.path.dirname(args.table)),
       "kernels": kern}
json.dump(out, open(args.out, "w"), indent=1)
for g, v in kern.items():
    print(g, v["hbm_bytes_per_launch"] / 1e6, "MB")
