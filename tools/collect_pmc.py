"""Summarise rocprofv3 --pmc passes (counter_collection CSVs under the given directories) for the dominant fused MVM
kernel into profiles/pmc_counters_current.json (read by bench.py for roofline.traffic).
usage: collect_pmc.py <out.json> <dir> [<dir> ...]"""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "randomly-projected-additive-gps_amd", "csrc")


def kernel_source_sha256(kernel_name):
    """sha256 over the source files the measured kernel is compiled from (bench.py recomputes it: a profile of another
    source gives roofline.traffic = null)."""
    files = ["rpgp_fact_asm.hip", "rpgp_fact_asm_loop.inc"] if kernel_name == "mvm_fact_asm_kernel" else ["rpgp_kernels.hip"]
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()

out, dirs = sys.argv[1], sys.argv[2:]
sums, counts, kname = {}, {}, None
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "mvm_fact_asm_kernel" in n:
                kname = "mvm_fact_asm_kernel"
            elif "mvm_fact_kernel<20, 1, 2>" in n:
                kname = "mvm_fact_kernel<20,1,2>"
            elif "mvm_tile_kernel<20, 1, 2, true" in n:
                kname = "mvm_tile_kernel<20,1,2,sym>"
            else:
                continue
            c = r["Counter_Name"]
            sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"])
            counts[c] = counts.get(c, 0) + 1
means = {c: sums[c] / counts[c] for c in sorted(sums)}
res = {"kernel": kname, "workload": "N=50000 J=20 T=1 (bench.py default, factorised prepared path)", "N": 50000, "J": 20,
       "T": 1, "fast": kname is not None and "fact" in kname,
       # ties this file to the kernel source it was measured on: bench.py reports roofline.traffic only when the sha256 of
       # the rpgp_kernels.hip it runs equals this one
       "kernel_source_sha256": kernel_source_sha256(kname), "launches_per_counter": counts, "per_launch_means": means,
       "notes": "rocprofv3 --pmc, separate passes (FETCH_SIZE / WRITE_SIZE / SQ+GRBM), gfx950. FETCH_SIZE and WRITE_SIZE are "
                "in KB. MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so hbm_bytes is "
                "bracketed as [(FETCH+WRITE)*1024, (2*FETCH+WRITE)*1024]."}
if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
    res["hbm_bytes_low"] = (means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024
    res["hbm_bytes_high"] = (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024
if "SQ_INSTS_VALU_TRANS_F32" in means and "SQ_INSTS_VALU" in means and "GRBM_GUI_ACTIVE" in means:
    trans, valu = means["SQ_INSTS_VALU_TRANS_F32"], means["SQ_INSTS_VALU"]
    other = valu - trans
    need = (trans * 8.2 + other * 4.4) / 1024.0          # SIMD-cycles by the measured issue costs (packed VALU: 4.4)
    if kname == "mvm_fact_asm_kernel":
        # the generated loop's own mix per step (tools/gen_fact_asm.py): 40 exp, 41 packed, 2 v_fmac (2.2), 1 DPP move (4.4)
        need = (trans * 8.2 + (other - trans * 2.0 / 40.0) * 4.4 + trans * 2.0 / 40.0 * 2.2) / 1024.0
    have = means["GRBM_GUI_ACTIVE"] / 8.0                 # the counter is summed over the 8 XCDs
    res["trans_wave_instr_expected"] = 50000.0 * 50000.0 * 20 / 2 / 64
    res["issue_model"] = "%.4g trans x 8.2 cyc + %.4g other VALU x 4.4 cyc = %.4g SIMD-cycles per SIMD; GRBM_GUI_ACTIVE/8 = " \
                         "%.4g cycles per launch -> %.1f%% of the issue bound" % (trans, other, need, have, 100.0 * need / have)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
