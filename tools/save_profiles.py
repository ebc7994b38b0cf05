"""Copy the rocprofv3 outputs of the last gpurun (gpurun_out/) into profiles/ as judged artefacts."""
import csv, glob, collections, json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r06"
pmc_only = "--pmc-only" in sys.argv


def kname(n):
    return n.split("(")[0].replace("void ", "").replace("dvbt::", "").split("<")[0]


out = {}
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "pmc", "*", "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    for k in acc:
        for c, v in acc[k].items():
            gmax = max(g for g, _ in v)                      # the timed steps' dispatches (the pre-scan's are smaller)
            big = [x for g, x in v if g == gmax]
            top = max(big)                                   # persistent kernels launch the same grid for the pre-scan's short stream: keep the full-size dispatches
            big = [x for x in big if x >= 0.5 * top]
            out.setdefault(k, {})[c] = sum(big) / len(big)
if pmc_only:
    for k, d in out.items():
        if d.get("GRBM_GUI_ACTIVE", 0) > 200000 or "viterbi" in k:
            print(k, {c: round(v, 1) for c, v in sorted(d.items())})
    sys.exit(0)
prof = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_final", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
shutil.copy(prof, os.path.join(root, "profiles", f"{tag}_kernel_stats_8k_qam64_7_8_65sf.csv"))                      # the default command (3 steps in flight)
solo = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_solo", "*kernel_stats.csv")), key=os.path.getmtime)
if solo:
    shutil.copy(solo[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats_8k_qam64_7_8_65sf_one_step_in_flight.csv"))   # bench.py --pipeline 1: the kernels' own durations
    prof = solo[-1]
if os.path.exists(os.path.join(root, "gpurun_out", "pipeline_depth.jsonl")):
    shutil.copy(os.path.join(root, "gpurun_out", "pipeline_depth.jsonl"), os.path.join(root, "profiles", f"{tag}_pipeline_depth.jsonl"))
# (gpurun_out/ outlives a round: what tools/final.sh wrote in an earlier one -- the soft-mode profile, the SNR sweep, the RS load series, the MFMA microbenchmark -- is not this round's)
fresh = lambda f: os.path.getmtime(f) > os.path.getmtime(os.path.join(root, "gpurun_out", "bench_final.json")) - 4 * 3600
soft = [f for f in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_soft", "*kernel_stats.csv")), key=os.path.getmtime) if fresh(f)]
if soft:
    shutil.copy(soft[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats_soft_8k_qam64_7_8_17sf.csv"))   # tools/soft_prof.py: the soft-decision chain
cfg5 = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_cfg5", "*kernel_stats.csv")), key=os.path.getmtime)
if cfg5:
    shutil.copy(cfg5[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats_config5_8dB.csv"))                  # tools/period_prof.py 8 16: BASELINE config 5 at the prescribed noise
for extra in ("config5_walk.jsonl", "snr_sweep.jsonl", "rs_load.jsonl", "ubench_mfma.json", "soft_gain.jsonl"):
    if os.path.exists(os.path.join(root, "gpurun_out", extra)) and os.path.getsize(os.path.join(root, "gpurun_out", extra)) > 0 and fresh(os.path.join(root, "gpurun_out", extra)):
        shutil.copy(os.path.join(root, "gpurun_out", extra), os.path.join(root, "profiles", f"{tag}_{extra}"))
cpp = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_cpp_lent", "*kernel_stats.csv")), key=os.path.getmtime)
if cpp:
    shutil.copy(cpp[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats_cpp_host.csv"))                     # tools/cpp_prof.py: the C++ multi-GPU host's bench mode, samples lent
cpc = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_cpp_lent", "*memory_copy_stats.csv")), key=os.path.getmtime)
if cpc:
    shutil.copy(cpc[-1], os.path.join(root, "profiles", f"{tag}_memory_copy_stats_cpp_host.csv"))                # ... and its copies by direction (no device-to-host copy of a slot: the runs stay on the device)
for src, dst in (("verify_ab.jsonl", "verify_ab_one_step_in_flight.jsonl"), ("bench_short.json", "bench_short.json")):
    if os.path.exists(os.path.join(root, "gpurun_out", src)) and os.path.getsize(os.path.join(root, "gpurun_out", src)) > 0 and fresh(os.path.join(root, "gpurun_out", src)):
        shutil.copy(os.path.join(root, "gpurun_out", src), os.path.join(root, "profiles", f"{tag}_{dst}"))
for src, dst in (("front_priority.jsonl", "front_priority.jsonl"), ("shard_scan.txt", "shard_scan.txt"), ("parity_sweep.txt", "parity_sweep.txt"), ("parity_sweep4.txt", "parity_sweep4.txt")):
    if os.path.exists(os.path.join(root, "gpurun_out", src)) and os.path.getsize(os.path.join(root, "gpurun_out", src)) > 0:
        shutil.copy(os.path.join(root, "gpurun_out", src), os.path.join(root, "profiles", f"{tag}_{dst}"))
sw = sorted(glob.glob(os.path.join(root, "gpurun_out", "parity_sweep5_*.txt")))
if sw:                                                                                                           # tools/sweep5.py, one file per seed: the streaming entry through lost locks
    with open(os.path.join(root, "profiles", f"{tag}_parity_sweep_stream.txt"), "w") as o:
        for f in sw:
            o.write("==== tools/sweep5.py 60 %s\n" % os.path.basename(f)[len("parity_sweep5_"):-4])
            o.write(open(f).read())
b = json.load(open(os.path.join(root, "gpurun_out", "bench_final.json")))
json.dump(b, open(os.path.join(root, "profiles", f"{tag}_bench_n1.json"), "w"), indent=1)
pb = json.load(open(os.path.join(root, "gpurun_out", "pmc", "FETCH_SIZE.json")))
summary = {"command": "tools/pmc.sh: rocprofv3 --pmc <group> --output-format csv -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-extras --superframes 64 (one pass per counter group: FETCH_SIZE | WRITE_SIZE | SQ issue counters | LDS counters); per kernel the average over its largest dispatches (the timed steps)",
           "workload": pb["config"],
           "note": "FETCH_SIZE/WRITE_SIZE in KB per dispatch. gfx950: FETCH_SIZE counts 64 B per 128 B request on coalesced streams, so it is doubled in hbm_bytes_corrected (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles.",
           "viterbi_algorithmic_bytes": pb["roofline"]["algorithmic_bytes_per_launch"], "kernels": {}}
if os.path.getmtime(os.path.join(root, "gpurun_out", "pmc", "FETCH_SIZE.json")) < os.path.getmtime(prof) - 3600:   # counters from an earlier run than the kernel stats
    summary["note"] += "  The PMC passes are older than the kernel stats beside them (tools/final_short.sh does not repeat them): small kernels may have changed since."
for k, c in out.items():
    e = dict(c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
    summary["kernels"][k] = e
json.dump(summary, open(os.path.join(root, "profiles", f"{tag}_pmc_summary_8k_qam64_7_8_65sf.json"), "w"), indent=1)
v = summary["kernels"]["viterbi3_kernel"]
print("viterbi3: hbm", v["hbm_bytes_corrected"] / 1e6, "MB; algorithmic", summary["viterbi_algorithmic_bytes"] / 1e6, "MB")
for r in list(csv.DictReader(open(prof)))[:16]:
    print(kname(r["Name"])[:30].ljust(30), r["Calls"].rjust(5), f"{float(r['AverageNs'])/1e3:10.1f} us", r["Percentage"])
