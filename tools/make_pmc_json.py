"""profiles/rNN_pmc_traffic.json (the file bench.py reads for `roofline.traffic`) from the per-kernel
PMC summaries that tools/prof_cmd.sh + tools/pmc_mean.py leave in gpurun_out/prof_bench/.
usage: python tools/make_pmc_json.py gpurun_out/prof_bench profiles/r02 "<build description>" """
import csv
import glob
import json
import os
import sys

src, dst, build = sys.argv[1], sys.argv[2], sys.argv[3]
tab = {}
for fn in sorted(glob.glob(os.path.join(src, "pmc*.csv"))):
    lines = open(fn).read().strip().split("\n")
    head = lines[0].split(",")
    rows = []
    for ln in lines[1:]:  # kernel names hold commas (template arguments): split from the right
        parts = ln.rsplit(",", len(head) - 1)
        rows.append(dict(zip(head, [parts[0].strip('"')] + parts[1:])))
    for row in rows:
        k = (row["kernel"], int(row["grid"]))
        e = tab.setdefault(k, {"launches": int(row["launches"])})
        for c, v in row.items():
            if c not in ("kernel", "grid", "launches") and v != "":
                e[c] = float(v)


def full(prefix):
    """the full-size launches (largest grid) of the kernels whose name starts with prefix"""
    ks = [k for k in tab if k[0].startswith(prefix)]
    g = max(k[1] for k in ks)
    return {k[0]: tab[k] for k in ks if k[1] == g}


def rw(e):  # FETCH_SIZE / WRITE_SIZE are KB of 1024 B; gfx950 reports 1/2 of wide coalesced reads
    r, w = e["FETCH_SIZE"] * 1024 * 2, e["WRITE_SIZE"] * 1024
    return {"read": r, "write": w, "hbm_bytes_per_launch": r + w, "launches": e["launches"]}


out = {"note": "rocprofv3 --kernel-trace --pmc <set>, one pass per line of tools/pmc_sets_bench.txt, of "
               "`python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none` (tools/prof_cmd.sh); mean over "
               "the full-size launches of a kernel (tools/pmc_mean.py); FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads), WRITE_SIZE as is, "
               "unit KB = 1024 B",
       "build": build,
       "config": {"N": 1000000, "d": 128, "dtype": "f64", "T": 32, "nq": 10000, "k": 10}}
wide = full("proj_mfma_wide")
w96 = [v for k, v in wide.items() if ",6,4,8," in k][0]
w128 = [v for k, v in wide.items() if ",8,8,8," in k][0]
n96, n128 = w96["launches"], w128["launches"]
a, b = rw(w96), rw(w128)
out["proj_mfma_wide"] = {
    "cols": (96 * n96 + 128 * n128) / (n96 + n128),
    "hbm_bytes_per_launch": (a["hbm_bytes_per_launch"] * n96 + b["hbm_bytes_per_launch"] * n128) / (n96 + n128),
    "read": (a["read"] * n96 + b["read"] * n128) / (n96 + n128),
    "write": (a["write"] * n96 + b["write"] * n128) / (n96 + n128),
    "96_hyperplanes": a, "128_hyperplanes": b,
    "mfma_util_pct": {"96_hyperplanes": w96["MfmaUtil"], "128_hyperplanes": w128["MfmaUtil"],
                      "source": "rocprofv3 --pmc MfmaUtil MfmaFlopsF64, %s_pmc.csv" % dst},
    "mfma_flops_f64_per_launch": {"96_hyperplanes": w96["MfmaFlopsF64"], "128_hyperplanes": w128["MfmaFlopsF64"]},
    "lds": {n: {"bank_conflict_cycles": w["SQ_LDS_BANK_CONFLICT"], "lds_active_cycles": w["SQ_LDS_IDX_ACTIVE"],
                "conflict_share": w["SQ_LDS_BANK_CONFLICT"] / w["SQ_LDS_IDX_ACTIVE"],
                "wait_inst_lds": w.get("SQ_WAIT_INST_LDS"), "wave_cycles": w.get("SQ_WAVE_CYCLES")}
            for n, w in (("96_hyperplanes", w96), ("128_hyperplanes", w128))},
    "note": "writes include the 16-bit codes of the streamed levels (10 of 13 levels: +19 % of the P bytes)"}
ex = full("proj_exact_lds")
if ex:
    e = list(ex.values())[0]
    out["proj_exact_lds"] = dict(rw(e), cols=32)
kn = full("knn_fused_kernel")
for name, e in kn.items():
    # template arguments: <TD, TK, PRE32, CSR, I8>
    key = ("int8_prefilter" if name.endswith(",true,false,true>") else
           "prefilter" if ",true,false" in name else "all_f64")
    out.setdefault("knn_fused_kernel", {})["%s (%s)" % (key, name)] = rw(e)
    if key in ("int8_prefilter", "prefilter") and ("knn_fused" not in out or key == "int8_prefilter"):
        out["knn_fused"] = {"hbm_bytes_per_launch": rw(e)["hbm_bytes_per_launch"], "kernel": name}
out["split"] = {}
for pre in ("wsub_kernel", "wsort_kernel", "wpack_kernel", "stream_assign", "stream_hist", "stream_to_perm", "stream_mid"):
    if not [k for k in tab if k[0].startswith(pre)]:
        continue
    for name, e in full(pre).items():
        out["split"][name] = rw(e)
json.dump(out, open(dst + "_pmc_traffic.json", "w"), indent=1)
# one merged per-kernel table of every counter
cols = sorted({c for e in tab.values() for c in e if c != "launches"})
with open(dst + "_pmc.csv", "w") as f:
    f.write("# %s\n# %s\n" % (build, out["note"]))
    f.write("kernel,grid,launches," + ",".join(cols) + "\n")
    for (k, g), e in sorted(tab.items()):
        f.write('"%s",%d,%d,' % (k, g, e["launches"]) + ",".join("%.5g" % e[c] if c in e else "" for c in cols) + "\n")
print(json.dumps(out["proj_mfma_wide"], indent=1)[:1500])
