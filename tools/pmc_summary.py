#!/usr/bin/env python3
"""profiles/pmc_summary.json from the raw per-kernel counter means (tools/pmc_parse.py output).
    python tools/pmc_summary.py <raw.json> <out.json> [note] [kmeans_raw.json]
kmeans_raw.json: the counters of `tools/kmeans_one.py 48` (2^20-row launches ONLY) — the k-means row is taken from there, because
the bench command runs the statistics kernel at 65 536 AND 2^20 rows and a mean over both sizes compares with nothing.
FETCH_SIZE / WRITE_SIZE are KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes of
wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is used as reported."""
import json
import sys

raw = json.load(open(sys.argv[1]))
note = sys.argv[3] if len(sys.argv) > 3 else ""
km_raw = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else None
B, M, K, NC, NQ, NB = 49152, 48, 256, 8841823, 1200, 1 << 20
ALG = {  # SURVEY 8(d) per-unit bytes x units per launch
    "sk_sweep_kernel": ("sk_sweep2_kernel<2, true>", B * M * K * 4),
    "adc_screen_q16_kernel": ("adc_screen_q16_kernel<48>", NQ * NC * M),
    "assign_mfma_kernel": ("assign_mfma_kernel<16>", NB * (768 * 4 + M)),
    "kmeans_stats_kernel": ("kmeans_stats_px_kernel<0>", NB * (768 * 4 + M)),   # launch A, 2^20-row launches only (kmeans_raw.json)
}
out = {"_how": "tools/pmc_collect.sh: rocprofv3 --pmc <group> --kernel-trace, one pass per counter group, over "
               "`python bench.py --steps 1 --warmup 1 --no-cpu --no-per-rank --no-opq --adc-batches 1`; means per launch. FETCH_SIZE/WRITE_SIZE are "
               "KiB; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md. " + note}
for key, (kname, alg) in ALG.items():
    src = raw
    if key == "kmeans_stats_kernel":
        if km_raw is None:
            continue                                       # no size-pure counters: no row (a mixed mean is worse than none)
        src = km_raw
    c = src.get(kname)
    if not c:                                              # template argument lists grow: match on the leading arguments
        hits = [k for k in src if k.startswith(kname.rstrip(">"))]
        if not hits:
            continue
        kname, c = hits[0], src[hits[0]]
    f, w = c.get("FETCH_SIZE", {}).get("mean"), c.get("WRITE_SIZE", {}).get("mean")
    e = {"kernel": kname, "launches": c.get("FETCH_SIZE", {}).get("n"), "fetch_size_kib_mean": f, "write_size_kib_mean": w,
         "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024) if f is not None and w is not None else None,
         "algorithmic_bytes_per_launch": alg,
         **({"source": "tools/kmeans_one.py 48: 2^20-row launches only"} if src is km_raw else {}),
         "sq": {k: v["mean"] for k, v in c.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")}}
    sq = e["sq"]
    if "SQ_INSTS_VALU" in sq and "GRBM_GUI_ACTIVE" in sq:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs
        e["valu_busy_frac"] = round(sq["SQ_INSTS_VALU"] * 4 / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    if sq.get("SQ_LDS_IDX_ACTIVE") and "GRBM_GUI_ACTIVE" in sq:
        # SQ_LDS_IDX_ACTIVE: LDS-array cycles summed over the 256 CUs; SQ_LDS_BANK_CONFLICT: the share that are conflict cycles
        e["lds_busy_frac"] = round(sq["SQ_LDS_IDX_ACTIVE"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 256), 3)
        e["lds_bank_conflict_frac"] = round(sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"], 3)
    if sq.get("SQ_VALU_MFMA_BUSY_CYCLES") and "GRBM_GUI_ACTIVE" in sq:
        # SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs (32 per 32x32x16 bf16 / 32x32x32 i8 MFMA)
        e["mfma_busy_frac"] = round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
    out[key] = e
json.dump(out, open(sys.argv[2], "w"), indent=1)
print({k: (v.get("hbm_bytes_per_launch"), v.get("valu_busy_frac")) for k, v in out.items() if k != "_how"})
