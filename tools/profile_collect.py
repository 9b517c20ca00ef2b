"""Turns what tools/profile_all.sh left under gpurun_out/<tag>/ into the committed files under profiles/:
  profiles/<tag>_bench_n1.json             the bench line of that run
  profiles/<tag>_rocprofv3_kernel_stats.md per-kernel totals of the rocprofv3 --kernel-trace --stats run
  profiles/pmc_traffic.json                HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) with the source fingerprint bench.py checks
  profiles/<tag>_pmc_mfma.json             MFMA / VALU busy counters of the update kernels (per launch) + derived utilisation
  profiles/<tag>_pmc_env_step.json         SQ counters of env_step_kernel (per launch) + derived figures
usage: python tools/profile_collect.py <tag>
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", tag)
    out = os.path.join(ROOT, "profiles")
    for name in ("bench_n1.json", "kernel_stats.md", "pmc_traffic.json", "pmc_bench_summary.json", "pmc_env_summary.json"):
        if "Traceback" in open(os.path.join(src, name)).read():
            raise SystemExit(f"profile_collect.py: {name} holds a traceback - refusing")
    line = [l for l in open(os.path.join(src, "bench_n1.json")) if l.startswith("{")][-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(out, f"{tag}_bench_n1.json"), "w"), indent=1)
    md = open(os.path.join(src, "kernel_stats.md")).read()
    with open(os.path.join(out, f"{tag}_rocprofv3_kernel_stats.md"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants ({tag}, one MI355X)\n\n" + md)
    # HBM traffic (passes 0 = FETCH_SIZE, 1 = WRITE_SIZE of the bench command)
    tr = json.load(open(os.path.join(src, "pmc_traffic.json")))      # written on the GPU box by tools/pmc_traffic.py (carries the source fingerprint)
    try:
        tr["_meta"]["git"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        pass
    json.dump(tr, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    # MFMA utilisation
    pm = json.load(open(os.path.join(src, "pmc_bench_summary.json")))
    keep = {}
    for k, v in pm.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) == 0:
            continue
        rec = {c: v[c] for c in sorted(v) if c.startswith("SQ_") or c in ("GRBM_GUI_ACTIVE", "launches")}
        # SQ_BUSY_CYCLES is summed over the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles)
        if v.get("SQ_BUSY_CYCLES"):
            rec["mfma_busy_fraction_of_simd_time"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["SQ_BUSY_CYCLES"] / 32 * 1024)
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_WAVE_CYCLES"):
            rec["valu_active_fraction_of_wave_time"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
        keep[k] = rec
    json.dump(dict(_about="tools/pmc_sq.py --set mfma on `bench.py --steps 1 --warmup 1`; means per launch; mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / "
                          "(SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs): the share of the launch's SIMD-cycles in which the matrix pipe was busy "
                          "(kernels overlap on 2-4 streams, so a launch's cycles include time it shares the chip)", kernels=keep),
              open(os.path.join(out, f"{tag}_pmc_mfma.json"), "w"), indent=1, sort_keys=True)
    # env step counters
    pe = json.load(open(os.path.join(src, "pmc_env_summary.json")))
    for k, v in pe.items():
        if "env_step" not in k:
            continue
        n = v.get("SQ_WAVES", 8192)
        d = dict(v)
        if v.get("SQ_BUSY_CYCLES"):
            cyc = v["SQ_BUSY_CYCLES"] / 32
            d["derived"] = dict(kernel_cycles=cyc, valu_instructions_per_env_step=v.get("SQ_INSTS_VALU", 0) / n, lds_instructions_per_env_step=v.get("SQ_INSTS_LDS", 0) / n,
                                vmem_reads_per_env_step=v.get("SQ_INSTS_VMEM_RD", 0) / n, vmem_writes_per_env_step=v.get("SQ_INSTS_VMEM_WR", 0) / n,
                                # issue cost of a wave64 vector instruction on a 16-lane SIMD: 4 cycles (v_add / v_fma ...), 8 for transcendentals
                                # (MI355X_MICROARCH.md, cycle constants; round 3 priced it at 2 and called an 85 %-busy kernel "42 % busy")
                                valu_issue_utilisation=((v.get("SQ_INSTS_VALU", 0) - v.get("SQ_INSTS_VALU_TRANS_F32", 0)) * 4 + v.get("SQ_INSTS_VALU_TRANS_F32", 0) * 8) / (1024 * cyc),
                                # independent check: SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave had a vector instruction executing
                                valu_active_utilisation=v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc),
                                fp32_arith_share_of_valu=(v.get("SQ_INSTS_VALU_ADD_F32", 0) + v.get("SQ_INSTS_VALU_MUL_F32", 0) + v.get("SQ_INSTS_VALU_FMA_F32", 0)) / max(v.get("SQ_INSTS_VALU", 1), 1),
                                mean_active_lanes_per_valu=v.get("SQ_THREAD_CYCLES_VALU", 0) / max(v.get("SQ_INSTS_VALU", 1), 1) if v.get("SQ_THREAD_CYCLES_VALU") else None,
                                wave_cycles_per_env_step=4 * v.get("SQ_WAVE_CYCLES", 0) / n, waiting_fraction=v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                                issue_stall_fraction=v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                                lds_bank_conflict_fraction=v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                                hbm_bytes_per_env_step=(2 * 1024 * v.get("FETCH_SIZE", 0) + 1024 * v.get("WRITE_SIZE", 0)) / n,
                                hbm_read_bytes_per_env_step=2 * 1024 * v.get("FETCH_SIZE", 0) / n, hbm_write_bytes_per_env_step=1024 * v.get("WRITE_SIZE", 0) / n)
        json.dump(dict(_about="tools/pmc_sq.py --set sq --set hbm on tools/bench_env.py 8192 (one control step of 8192 envs per launch); means per launch. "
                              "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves, SQ_BUSY_CYCLES is summed over 32 shader engines, "
                              "FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE doubled for bytes, gfx950)", counters=d),
                  open(os.path.join(out, f"{tag}_pmc_env_step.json"), "w"), indent=1, sort_keys=True)
    for name in ("bench_env.txt", "env_stamps.txt", "bench_ppo.txt"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            txt = "".join(l for l in open(p) if "amdgpu.ids" not in l)
            if "Traceback" in txt or not txt.strip():      # a failed tool run is not a profile (round 2 committed one)
                raise SystemExit(f"profile_collect.py: {p} holds a traceback or nothing - refusing to commit it as evidence")
            open(os.path.join(out, f"{tag}_{name}"), "w").write(txt)
    print(json.dumps({k: bench[k] for k in ("value", "ms_per_step")}))


if __name__ == "__main__":
    main()
