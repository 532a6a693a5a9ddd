#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.txt (profiles/summarize_pmc.py) -> profiles/pmc_traffic.json: HBM bytes per launch of the kernels bench.py prices.
FETCH_SIZE (KB) is doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE (KB) is used
as reported.  Usage: python profiles/pmc_to_traffic.py profiles/r04_pmc_summary.txt r04"""
import json, os, re, sys
src, rnd = sys.argv[1], sys.argv[2]
blocks = {}
cur = None
for line in open(src):
    if line.startswith("---"):
        break
    if not line.startswith(" "):
        cur = line.strip()
        blocks[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", line)
        if m and cur:
            blocks[cur][m.group(1)] = float(m.group(3))


def traffic(name):
    b = blocks[name]
    return {"bytes": 2 * b["FETCH_SIZE"] * 1024 + b["WRITE_SIZE"] * 1024, "FETCH_SIZE_KB": b["FETCH_SIZE"], "WRITE_SIZE_KB": b["WRITE_SIZE"]}


def find(prefix):
    return [k for k in blocks if k.startswith(prefix)]


fwd, bwd = traffic("render_fwd2x_k<1, 1, 1, 1, 1, 8, false, false>"), traffic("render_bwd_pair_k<1, 1, 1, 1, 1, false, false, false, 32>")
fr, br = traffic("render_fwd_reg_k<1, 1, 1, 1, 1, false, false>"), traffic("render_bwd_pair_k<1, 1, 1, 1, 1, false, true, false, 32>")
out = {
    "_note": "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only). FETCH_SIZE (KB) is doubled per "
             "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for wide coalesced reads); WRITE_SIZE (KB) is used as reported (uncalibrated per the guide).",
    "_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/run_profiles_{rnd}.sh, {os.path.basename(src)}",
    "D32_T50_720x1280_mpv_v0": {"fwd": fwd["bytes"], "bwd": bwd["bytes"], "raw": {"fwd": fwd, "bwd": bwd},
                                "source": "render_fwd2x_k and render_bwd_pair_k<..., false> (the backward's fetch includes the owner table)"},
    "D32_T50_720x1280_stack792x1408_reg": {"fwd": fr["bytes"], "bwd": br["bytes"],
                                            "source": "render_fwd_reg_k and render_bwd_pair_k<..., true> on the 1.1x stack with the smoothness regularisers"},
}
f16f, f16b = find("render_fwd2x_k<1, 1, 1, 1, 1, 8, true, false>"), find("render_bwd_pair_k<1, 1, 1, 1, 1, true, false, false, 32>")
if f16f and f16b:
    a, b = traffic(f16f[0]), traffic(f16b[0])
    out["D32_T50_720x1280_fp16_stack"] = {"fwd": a["bytes"], "bwd": b["bytes"], "algorithmic": {"fwd": 50 * 720 * 1280 * (8 * 32 + 12), "bwd": 50 * 720 * 1280 * (16 * 32 + 12)},
                                           "source": "render_fwd2x_k<F16> and render_bwd_pair_k<F16>: cfg5's storage format on the cfg3 geometry"}
loss = {}
for k in blocks:
    if any(k.startswith(p) for p in ("patchnn6_k", "vote_fold_lds_k", "video_to_gram16_k")) and "FETCH_SIZE" in blocks[k]:
        loss[k] = traffic(k)["bytes"]
out["loss_720p_kernels"] = {"bytes_per_launch": loss, "compulsory_bytes_per_iteration": 4.0 * 720 * 1280 * (3 * 52 * 3 + 3 * 75 * 2),
                            "note": "one looping-loss iteration = video_to_gram16_k<false> (x) + patchnn6_k + vote_fold_lds_k (y prepared once per clip: "
                                    "video_to_gram16_k<true> is not part of an iteration); patchnn6_k<..., 16> is the ref-view configuration, <..., 14> the other views'"}
json.dump(out, open(os.path.join(os.path.dirname(src), "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1)[:1800])
