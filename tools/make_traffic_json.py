#!/usr/bin/env python3
"""Copies the summaries of tools/profile_round3.sh from gpurun_out/prof_<tag>/ into profiles/<tag>_* and derives the
HBM-traffic ratios bench.py quotes:

    profiles/traffic.json          dense leg: (2 x FETCH_SIZE + WRITE_SIZE) of k_xpass_dense / its algorithmic bytes
    profiles/traffic_sparse.json   sparse leg: the same per kernel (row pass, column pass, step phases) and for the whole tick

FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the counter tallies
128-byte requests at 64 bytes for wide coalesced reads. That calibration was made on 16-byte-per-lane streaming reads
(the dense kernel); the sparse kernels read 8 bytes per lane (packs) and 16 (staging) -- uncalibrated widths, so both the
raw and the doubled figure are recorded there.

    python tools/make_traffic_json.py r2
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_sums(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+%s\s+n=\s*(\d+)\s+sum=\s*([0-9.]+)" % counter, line)
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)) * 1024.0)       # KB -> bytes
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(src)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, "%s_%s" % (tag, f)))
    last = lambda p: json.loads(open(os.path.join(src, p)).read().strip().splitlines()[-1])
    # ---- dense
    bd = last("bench_dense_short.json")
    fe, wr = pmc_sums(os.path.join(src, "dense_pmc_fetch.txt"), "FETCH_SIZE"), pmc_sums(os.path.join(src, "dense_pmc_write.txt"), "WRITE_SIZE")
    kd = next(k for k in fe if "k_xpass_dense" in k)
    hbm = 2.0 * fe[kd][1] + wr[kd][1]
    alg = bd["all_launches"]["alg_bytes"]
    json.dump({"kernel": "k_xpass_dense<4,4>", "command": "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --loglik-iters 0 --no-sparse --no-sweep --no-profile",
               "launches": fe[kd][0], "FETCH_SIZE_bytes": fe[kd][1], "WRITE_SIZE_bytes": wr[kd][1],
               "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads; MI355X_MICROARCH.md, HBM section)",
               "hbm_bytes": hbm, "alg_bytes_same_launches": alg, "hbm_bytes_per_alg_byte": round(hbm / alg, 4),
               "source": ["profiles/%s_dense_pmc_fetch.txt" % tag, "profiles/%s_dense_pmc_write.txt" % tag]},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    # ---- sparse
    bs = last("bench_sparse_only.json")
    fe, wr = pmc_sums(os.path.join(src, "sparse_pmc_fetch.txt"), "FETCH_SIZE"), pmc_sums(os.path.join(src, "sparse_pmc_write.txt"), "WRITE_SIZE")
    alg = bs["all_launches"]["alg_bytes_row_plus_column"]
    per, tot_raw, tot_x2 = {}, 0.0, 0.0
    for keys, label in ((("k_rowpass_lds", "k_rowcold"), "row pass"), (("k_colpass_lds",), "column pass"), (("k_step_a",), "step A"),
                        (("k_step_b",), "step B"), (("k_step_c(",), "step C"), (("k_step_commit",), "step commit")):
        kf = [k for k in fe if any(key in k for key in keys)]
        kw = [k for k in wr if any(key in k for key in keys)]
        f = sum(fe[k][1] for k in kf)
        w = sum(wr[k][1] for k in kw)
        n = sum(fe[k][0] for k in kf if keys[0] in k)        # (the cold-slice launch in front of the row kernel is part of the row pass)
        per[label] = {"launches": n, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w, "hbm_bytes_raw": f + w, "hbm_bytes_fetch_x2": 2 * f + w}
        tot_raw += f + w
        tot_x2 += 2 * f + w
    half = alg / 2.0
    for label in ("row pass", "column pass"):
        per[label]["alg_bytes"] = half
        per[label]["raw_per_alg_byte"] = round(per[label]["hbm_bytes_raw"] / half, 4)
        per[label]["fetch_x2_per_alg_byte"] = round(per[label]["hbm_bytes_fetch_x2"] / half, 4)
    json.dump({"command": "python bench.py --sparse-only --sparse-cpu-sample 0", "workload": bs["workload"],
               "alg_bytes_row_plus_column_all_launches": alg,
               "note": "algorithmic bytes = SURVEY 8(d) B_pass = nnz*4 + 8l + 8n per pass and active problem (int32 ids); the kernels read "
                       "uint16 ids, so the passes can move fewer bytes than that; the step's n-vector streams count as zero algorithmic bytes",
               "hbm_bytes_per_alg_byte": {"raw": round(tot_raw / alg, 4), "fetch_x2": round(tot_x2 / alg, 4)},
               "per_kernel": per, "source": ["profiles/%s_sparse_pmc_fetch.txt" % tag, "profiles/%s_sparse_pmc_write.txt" % tag]},
              open(os.path.join(dst, "traffic_sparse.json"), "w"), indent=1)
    print(open(os.path.join(dst, "traffic.json")).read())
    print(open(os.path.join(dst, "traffic_sparse.json")).read())


if __name__ == "__main__":
    main()
