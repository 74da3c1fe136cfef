#!/usr/bin/env python3
"""Copy the round's evidence from gpurun_out/ (scratch) into profiles/ (tracked), stamped with the commit and the hash of csrc/ it was taken on.
Inputs: gpurun_out/fin3 (tools/gpu_round3_final1.sh), gpurun_out/fin3b (…final2.sh), gpurun_out/prof3 (tools/profile_round.sh [+ tools/infer_bench.py > prof3/infer.txt]).
Run tools/pmc_report.py first (it writes r03_mfma_busy.json / r03_hbm_traffic.json)."""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
import bench
sha = bench.csrc_sha16()
head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD']).decode().strip()
P = 'profiles/'
NOISE = '/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n'
for src, dst in (('gpurun_out/prof3/stats', 'r03_bench_kernel_stats.csv'), ('gpurun_out/prof3/cheby', 'r03_cheby_alexnet_kernel_stats.csv'),
                 ('gpurun_out/prof3/fk', 'r03_fastkan_layer_kernel_stats.csv')):
    shutil.copy(glob.glob(src + '/**/*kernel_stats.csv', recursive=True)[0], P + dst)
shutil.copy('gpurun_out/prof3/bench_default.json', P + 'r03_bench_default.json')
for tag in 'ABFW':
    rows = []
    for fn in glob.glob(f'gpurun_out/prof3/{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"(k_(?:conv|band)_[a-z_]+(?:<[^>]*>)?)", r["Kernel_Name"])
            if m:
                rows.append((m.group(1), r["Counter_Name"], r["Counter_Value"], r["Start_Timestamp"], r["End_Timestamp"], r.get("Grid_Size", ""),
                             r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
    with open(P + f'r03_pmc_pass{tag}_conv_kernels.csv', 'w', newline='') as f:
        w = csv.writer(f); w.writerow(["kernel", "counter", "value", "start_ns", "end_ns", "grid", "lds_block", "vgprs"]); w.writerows(rows)
H = f"# round 3, commit {head} (csrc {sha}), MI355X, "
def put(src, dst, header):
    open(P + dst, 'w').write(header + open(src).read().replace(NOISE, ''))
put('gpurun_out/fin3b/noise_probe.txt', 'r03_noise_probe.txt', H + "python tests/noise_probe.py\n")
put('gpurun_out/fin3b/split_bf16_probe.txt', 'r03_split_bf16_probe.txt', H + "tools/probe/split_bf16_probe\n")
open(P + 'r03_split_bf16_conv.txt', 'w').write(H + "tools/probe/split_bf16_conv 20 ; the -DX_STAMP build (s_memtime / s_memrealtime around the main loop) 200 launches\n"
                                               + open('gpurun_out/fin3b/split_bf16_conv.txt').read() + open('gpurun_out/fin3b/split_bf16_conv_stamp.txt').read().split('\n')[0] + '\n')
put('gpurun_out/fin3b/ab_lpt.txt', 'r03_ab_pm_lpt.txt', H + "python tools/ab_env.py KAN_PM_LPT (the -DKAN_TUNING_KNOBS build, alternating arms on one box; off = launch order of round 2)\n")
fb = open('gpurun_out/fin3b/family_bench.txt').read().replace(NOISE, '')
open(P + 'r03_family_bench.txt', 'w').write(
    "# python tools/family_bench.py  (KAN-VGG11 built with every CONV_KAN_FACTORY family at its factory defaults as models/kan_vgg.py passes them; fwd + CE loss + bwd,\n"
    f"# bs 256, 3x32x32, MI355X, median of three 10-step windows after 3 warm-up steps; round 3, commit {head}, csrc {sha}).  TF = dense conv FLOPs at P planes / step time; Wav-KAN's\n"
    "# wavelet stage is not a GEMM (its figure counts only the two plain convolutions as P = 2).  Round 2: profiles/r02_family_bench.txt.\n" + fb)
d = json.load(open('gpurun_out/fin3b/exact_ab.json')); d['commit'] = head; d['csrc_sha16'] = sha
json.dump(d, open(P + 'r03_exact_transcendentals_ab.json', 'w'), indent=1)
rep = [tuple(map(float, l.split())) for l in open('gpurun_out/fin3b/repeats.txt') if l.strip()]
json.dump({"commit": head, "csrc_sha16": sha, "command": "python bench.py --no-cpu-baseline --no-aux (5 runs back to back on one box)",
           "images_per_sec": [r[0] for r in rep], "ms_per_step": [r[1] for r in rep]}, open(P + 'r03_bench_repeats.json', 'w'), indent=1)
t = open('gpurun_out/fin3/tests.txt').read().splitlines()
open(P + 'r03_gpu_tests.txt', 'w').write(H + "python -m pytest tests -q -m gpu --durations=8 ; python -c 'import __graft_entry__ as g; g.smoke()'\n" + "\n".join(t[-14:]) + "\n"
                                         + open('gpurun_out/fin3/smoke.txt').read())
if os.path.exists('gpurun_out/prof3/infer.txt'):
    open(P + 'r03_infer_bench.txt', 'w').write("# python tools/infer_bench.py (round 3, MI355X): KAN-VGG11 eval / no_grad, bs 256; the last entry is the OPT-IN split-precision inference mode "
                                               "(DESIGN section 10)\n" + open('gpurun_out/prof3/infer.txt').read().splitlines()[-1] + "\n")
d = json.load(open(P + 'r03_bench_default.json'))
print(sha, head, d['value'], d['ms_per_step'], d['roofline']['executed_frac'], d['roofline']['end_to_end_executed_frac'])
