"""HBM traffic per Newton step and kernel family from rocprofv3 PMC passes of bench.py with 1 and with 3 timed steps (differenced:
set-up, warm-up and the per-kernel pass behind the timed region cancel), FETCH_SIZE and WRITE_SIZE collected in separate runs.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte requests at 64 bytes).

    python tools/pmc_families.py <fetch_1.csv> <fetch_3.csv> <write_1.csv> <write_3.csv> <out.json> [<n> = 128]
The CSVs are tools/rocpd_summary.py's <prefix>_pmc.csv (kernel, counter, dispatches, mean_per_dispatch, sum, ...).  Writes / merges
{"families": {"<n>": {family: {"traffic_bytes", "fetch_bytes", "write_bytes", "dispatches_per_step", "source"}}}} into out.json
(profiles/pmc_traffic.json: bench.py quotes it next to every kernel family)."""
import csv
import json
import re
import sys

FAMILIES = [  # first match wins
    ("assemble_cells", r"k_ins_assemble[23]"),
    ("spmv_uu", r"k_spmv_uu"),
    ("mf_cell", r"k_apply_uu_mf2"),
    ("mf_gather", r"k_mf_gather"),
    ("spmv_b_bt", r"k_spmv_planar<1, 3,|k_spmv_planar<3, 1,|k_spmv_planar_add"),
    ("spmv_sm", r"k_spmv_planar<1, 1, (32|64), float"),
    ("spmv_mp", r"k_spmv_planar<1, 1, 8"),
    ("mdot", r"k_mdot|k_reduce_final"),
    ("maxpy", r"k_maxpy"),
    ("mg_transfer", r"k_mg_csr|k_mg_inject|k_mg_mask"),
    ("smoother_setup", r"k_uu_diag|k_block_invert|k_bjac_setup"),
    ("cg_recurrence", r"k_cgd_|k_cg1_"),
    ("schur_setup", r"k_schur|k_mask_b"),
    ("zero_fill", r"fillBufferAligned"),
    ("vector_ops", r"k_axpy|k_axpby|k_scale|k_cheb|k_cvt|copyBuffer|k_mul|k_div|k_bjac_apply|k_vec_|k_recip|k_to_f32"),
]


def family(name):
    for fam, pat in FAMILIES:
        if re.search(pat, name):
            return fam
    return "other"


def load(path):
    d = {}
    for r in list(csv.reader(open(path)))[1:]:
        d[r[0]] = (int(r[2]), float(r[4]))  # dispatches, sum (KiB)
    return d


def main(f1, f3, w1, w3, out, n="128"):
    per = {}
    for which, a, b in (("fetch", load(f1), load(f3)), ("write", load(w1), load(w3))):
        for k, (c3, s3) in b.items():
            c1, s1 = a.get(k, (0, 0.0))
            if c3 <= c1:
                continue
            e = per.setdefault(family(k), {"fetch": 0.0, "write": 0.0, "disp": 0.0})
            e[which] += (s3 - s1) / 2 * 1024.0 * (2.0 if which == "fetch" else 1.0)
            if which == "fetch":
                e["disp"] += (c3 - c1) / 2
    fam = {k: {"traffic_bytes": v["fetch"] + v["write"], "fetch_bytes": v["fetch"], "write_bytes": v["write"], "dispatches_per_step": v["disp"],
               "source": "profiles/pmc_traffic.json <- tools/prof_families.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, runs with 1 and 3 timed steps differenced)"}
           for k, v in per.items()}
    try:
        doc = json.load(open(out))
    except (OSError, ValueError):
        doc = {}
    doc.setdefault("families", {})[str(n)] = fam
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["traffic_bytes"]):
        print(f"{k:16s} {v['traffic_bytes'] / 1e9:9.2f} GB per step  (fetched {v['fetch_bytes'] / 1e9:8.2f}, written {v['write_bytes'] / 1e9:8.2f}; {v['dispatches_per_step']:.0f} dispatches)")


if __name__ == "__main__":
    main(*sys.argv[1:])
