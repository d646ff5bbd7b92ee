#!/usr/bin/env python3
"""tools/pmc.py -- hardware counters of ONE kernel of a command, collected the way MI355X_MICROARCH.md prescribes:
rocprofv3 --kernel-trace --pmc, every counter set in its own pass (FETCH_SIZE and WRITE_SIZE each alone), nothing but
the kernel-trace domain next to --pmc, the program itself after `--`.  Sums the counters over the dispatches of the
kernel whose name contains KERNEL and writes one JSON summary (profiles/<tag>_pmc_summary.json) with the derived
figures the judge asks for: wait / VALU-busy shares, LDS bank conflicts, MFMA busy share and int8 MFMA rate against the
gfx950 dense peak, HBM bytes (FETCH_SIZE doubled per the gfx950 note) per launch.

    python3 tools/pmc.py --tag r02_mix --kernel k_intra_packed [--sets inst,wait,mfma,lds,traffic] -- python3 bench.py --frames 1728 --steps 1 --warmup 0 --no-cpu-baseline
Run on the GPU box (gpurun); scratch goes to gpurun_out/pmc_<tag>_<set>/.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {
    "inst": "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT",
    "wait": "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS",
    "mfma": "SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU",
    "lds": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES",
    # the vector-memory pipe (texture addresser, L1): busy against GRBM_GUI_ACTIVE x 256 CUs, and who stalls whom
    "mem1": "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum",
    "mem2": "TA_FLAT_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum",
    "mem3": "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum",
    "mem4": "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
}
INT8_DENSE_PEAK_TOPS = 5000.0  # MI355X dense int8 = the dense fp8 figure of the guides (~5 PFLOP/s); never the 2:1-sparsity number


def run_set(tag, name, counters, cmd, kernel, timeout):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{name}")
    shutil.rmtree(out, ignore_errors=True)
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--kernel-trace", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out, "--"] + cmd
    log = open(out + ".log", "w")
    rc = subprocess.call(full, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, timeout=timeout)
    files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
    if rc != 0 or not files:
        return None, rc
    tot, disp, dur = collections.Counter(), set(), {}
    for r in csv.DictReader(open(files[0])):
        if kernel in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    kt = glob.glob(os.path.join(out, "*", "*kernel_trace.csv"))
    if kt:
        for r in csv.DictReader(open(kt[0])):
            if kernel in r["Kernel_Name"]:
                dur[r.get("Dispatch_Id", len(dur))] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {"counters": {k: v for k, v in tot.items()}, "launches": len(disp), "kernel_ns": sum(dur.values())}, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kernel", required=True, help="substring of the kernel name")
    ap.add_argument("--sets", default="inst,wait,mfma,lds,fetch,write")
    ap.add_argument("--timeout", type=int, default=500)
    ap.add_argument("--meta", default="{}", help="JSON merged into the summary (frames, plans, ...)")
    ap.add_argument("--issue-cycles", type=float, default=3.55,
                    help="SIMD cycles per VALU instruction of the kernel's mix (default: k_intra_packed's static mix, 36 %% of its VALU "
                         "instructions in the 2.4-cycle class, 64 %% in the 4.2-cycle class)")
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        raise SystemExit("no command after --")
    summary = {"tag": a.tag, "kernel": a.kernel, "command": " ".join(cmd), "passes": {},
               "method": "rocprofv3 --kernel-trace --pmc, one pass per counter set, summed over the kernel's dispatches"}
    summary.update(json.loads(a.meta))
    C = {}
    for name in a.sets.split(","):
        res, rc = run_set(a.tag, name, SETS[name], cmd, a.kernel, a.timeout)
        if res is None:
            summary["passes"][name] = {"error": f"rocprofv3 exit {rc} or no counter file"}
            print(f"[pmc] set {name}: failed (rc {rc})", flush=True)
            continue
        summary["passes"][name] = res
        C.update(res["counters"])
        summary["launches"] = res["launches"]
        print(f"[pmc] set {name}: {res['launches']} launches, {res['kernel_ns'] / 1e6:.3f} ms in kernel", flush=True)

    def ratio(a_, b_):
        return round(C[a_] / C[b_], 5) if a_ in C and C.get(b_) else None

    d = {}
    d["wait_share_of_wave_cycles"] = ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
    d["wait_inst_share_of_wave_cycles"] = ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
    # VALU utilisation of the chip: a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs; the
    # clock is taken as 2.4 GHz (the counters carry no clock), the kernel time is the trace's
    ns_inst = summary["passes"].get("inst", {}).get("kernel_ns") or 0
    # VALU issue: measured on gfx950 with 2..8 waves per SIMD (tools/issue_probe.hip, profiles/r03_issue_probe.txt): plain VOP2 integer /
    # logic / shift-right / move / fp32 add-mul-fma instructions issue every ~2.4 SIMD cycles, everything else (all multiplies, VOP3
    # three-operand forms, min / max / med3, bit-field ops, left shifts, DPP, SGPR-sourced operands, packed 16-bit, dot) every ~4.2;
    # a wave alone on its SIMD ~5.5.  `valu_utilisation` keeps round 2's 4-cycle convention; the bracket below is what the mix allows.
    if ns_inst and "SQ_INSTS_VALU" in C:
        per = C["SQ_INSTS_VALU"] / (ns_inst * 2.4 * 1024)
        d["valu_utilisation"] = round(per * 4, 4)
        d["valu_utilisation_if_all_fast_2.4cyc"] = round(per * 2.4, 4)
        d["valu_utilisation_if_all_slow_4.2cyc"] = round(per * 4.2, 4)
        d["valu_utilisation_weighted"] = round(per * a.issue_cycles, 4)
        d["valu_issue_cycles_assumed"] = a.issue_cycles
    else:
        d["valu_utilisation"] = None
    d["valu_share_of_wave_cycles"] = ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
    d["valu_insts_per_wave"] = ratio("SQ_INSTS_VALU", "SQ_WAVES")
    d["salu_insts_per_wave"] = ratio("SQ_INSTS_SALU", "SQ_WAVES")
    d["lds_insts_per_wave"] = ratio("SQ_INSTS_LDS", "SQ_WAVES")
    d["lds_bank_conflict_share"] = ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
    d["mfma_busy_share_of_busy_cycles"] = ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES")
    if "SQ_INSTS_VALU_MFMA_I8" in C:
        ns = summary["passes"].get("mfma", {}).get("kernel_ns") or 0
        ops = C["SQ_INSTS_VALU_MFMA_I8"] * 2 * 32 * 32 * 32  # v_mfma_i32_32x32x32_i8: 32x32x32 multiply-adds per wave instruction
        d["mfma_i8_instructions"] = int(C["SQ_INSTS_VALU_MFMA_I8"])
        if ns:
            d["mfma_i8_tops"] = round(ops / ns / 1e3, 3)
            d["mfma_i8_share_of_dense_peak"] = round(ops / ns / 1e3 / INT8_DENSE_PEAK_TOPS, 6)
            d["int8_dense_peak_tops"] = INT8_DENSE_PEAK_TOPS
    if "FETCH_SIZE" in C and "WRITE_SIZE" in C and summary.get("launches"):
        n = summary["launches"]
        d["FETCH_SIZE_KB"], d["WRITE_SIZE_KB"] = C["FETCH_SIZE"], C["WRITE_SIZE"]
        d["hbm_bytes_per_launch"] = round((2 * C["FETCH_SIZE"] + C["WRITE_SIZE"]) * 1024 / n)
        d["hbm_read_bytes_per_launch"] = round(2 * C["FETCH_SIZE"] * 1024 / n)
        d["hbm_write_bytes_per_launch"] = round(C["WRITE_SIZE"] * 1024 / n)
        d["note_fetch"] = "gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM)"
    if "TA_TA_BUSY_sum" in C and C.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs' dispatches of the counter file: per-XCD cycles x 8; a TA per CU (256)
        d["ta_busy_share"] = round(C["TA_TA_BUSY_sum"] / (C["GRBM_GUI_ACTIVE"] / 8 * 256), 4)
        d["ta_busy_avr"] = C.get("TA_BUSY_avr")
    for k in ("TA_FLAT_WAVEFRONTS_sum", "TA_BUFFER_WAVEFRONTS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum",
              "TCP_GATE_EN1_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum",
              "TD_TD_BUSY_sum", "TD_TC_STALL_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TA_TCP_STATE_READ_sum", "GRBM_GUI_ACTIVE", "TA_TA_BUSY_sum",
              "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
        if k in C:
            d[k] = C[k]
    summary["derived"] = d
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"{a.tag}_pmc_summary.json")
    json.dump(summary, open(path, "w"), indent=1)
    print(json.dumps(d), flush=True)
    print("[pmc] wrote", path, "(copy it into profiles/)", flush=True)


if __name__ == "__main__":
    main()
