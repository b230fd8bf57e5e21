"""Summarise rocprofv3 --pmc counter_collection.csv files of the headline bench per k_step wavefront / launch.
    python tools/pmc_summary.py <dir> ...          one line per counter file
    python tools/pmc_summary.py --json <dir>       the profiles/r01_pmc_*.json document (all pmc_* sub-directories of <dir>)"""
import collections, csv, glob, json, os, sys


def collect(d):
    acc = collections.defaultdict(list); waves = 256
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].split("(")[0].strip() == "k_step":      # the velocity-drive step kernel (k_step_pd, k_step_dr are others)
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); waves = int(r["Grid_Size"]) // 64
    return {k: sum(v) / len(v) / waves for k, v in acc.items()}, waves


if __name__ == "__main__":
    if sys.argv[1] == "--json":
        per_wave, waves = collect(sys.argv[2])
        pw = {}
        for k, v in per_wave.items():
            if k in ("FETCH_SIZE", "WRITE_SIZE"): pw[k + "_KB"] = round(v, 2)
            elif k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY"): pw[k + "_quad"] = round(v, 1)
            else: pw[k] = round(v, 1)
        # FETCH_SIZE reads half of the bytes on gfx950 (MI355X_MICROARCH.md, HBM section) - for one dword per lane as well as for 16 B per lane
        # (profiles/r02_fetch_calib.json, tools/microbench/fetch_calib.hip); WRITE_SIZE reads the bytes exactly
        fetch = per_wave.get("FETCH_SIZE", 0.0) * 2048 * waves; write = per_wave.get("WRITE_SIZE", 0.0) * 1024 * waves
        # fp32 VALU instruction counters: a packed instruction (v_pk_fma_f32 ...) does two lanes' worth of arithmetic per lane but is ONE
        # instruction, so this is a LOWER bound of the flops since round 2 (the step kernel issues about a third of its fp32 work packed)
        flop = (per_wave.get("SQ_INSTS_VALU_ADD_F32", 0) + per_wave.get("SQ_INSTS_VALU_MUL_F32", 0) + per_wave.get("SQ_INSTS_VALU_TRANS_F32", 0)
                + 2 * per_wave.get("SQ_INSTS_VALU_FMA_F32", 0)) * 64 / 16          # per-lane ops of a 64-lane wavefront holding 16 envs
        wc, wa, vi = per_wave.get("SQ_WAVE_CYCLES", 0.0), per_wave.get("SQ_WAIT_ANY", 0.0), per_wave.get("SQ_INSTS_VALU", 0.0)
        print(json.dumps({
            "source": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), bench.py --steps 100 --warmup 20, k_step dispatches only, "
                      f"4096 envs = {waves} wavefronts per launch (tools/profile_round.sh)",
            "per_wave": pw,
            "per_launch": {"fetch_bytes": fetch, "write_bytes": write, "hbm_traffic_bytes": fetch + write,
                           "note": "fetch_bytes = FETCH_SIZE x 2 (the gfx950 correction, checked for 4 B/lane loads in profiles/r02_fetch_calib.json); "
                                   "the bench consumes the clipped out_* buffers only, so the engine's unclipped obs_buf / states_buf / reward-term copies (+672 B/env, written once lm_ptr() has handed them out) are not in these writes; algorithmic bytes: 1488 B/env-step x the envs of a launch"},
            "wave": {"valu_instructions": round(vi, 1), "wave_quad_cycles": round(wc, 1), "wait_quad_cycles": round(wa, 1),
                     "valu_issue_fraction": round(vi / wc, 4) if wc else None, "wait_fraction": round(wa / wc, 4) if wc else None,
                     "note": "one wavefront per SIMD: a VALU instruction occupies one quad-cycle issue slot, so valu_issue_fraction is the share of the "
                             "wavefront's lifetime spent issuing vector arithmetic"},
            "flop_per_env_step_lower_bound": flop,
            "flop_note": "packed fp32 instructions are counted once by SQ_INSTS_VALU_*_F32"}, indent=1))
    else:
        for d in sys.argv[1:]:
            pw, waves = collect(d)
            print(d, {k: round(v, 1) for k, v in pw.items()})
