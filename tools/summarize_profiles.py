"""Turn the rocprofv3 outputs a GPU run left under gpurun_out/ into the small summaries committed in profiles/.

    python tools/summarize_profiles.py <round-tag>       e.g.  r1

Inputs (produced on the GPU box, see profiles/README.md for the exact commands):
  gpurun_out/prof_<tag>/**/_kernel_stats.csv        rocprofv3 --kernel-trace --stats      -- python3 bench.py ...
  gpurun_out/pmc_fetch/**/_counter_collection.csv   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py ...
  gpurun_out/pmc_write/**/_counter_collection.csv   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py ...
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports exactly half the bytes of a 16-B-per-lane coalesced read stream (MI355X_MICROARCH.md §HBM),
which is what the patch kernel's window loads are.
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys


def newest(files):
    return sorted(files, key=os.path.getmtime)[-1:]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
    out = os.path.join(ROOT, 'profiles')
    os.makedirs(out, exist_ok=True)
    stats = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof_' + tag, '**', '*_kernel_stats.csv'), recursive=True))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out, tag + '_kernel_stats.csv'), 'w', newline='') as f:
            csv.writer(f).writerows(rows[:12])
    pmc = {}
    for name, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        files = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', name, '**', '*_counter_collection.csv'), recursive=True))
        if not files:
            continue
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r['Counter_Name'] == ctr and 'dmf::' in r['Kernel_Name']:
                d[r['Kernel_Name']].append(float(r['Counter_Value']))
        for k, v in d.items():
            pmc.setdefault(k, {})[ctr] = {'n': len(v), 'median_KiB': statistics.median(v), 'mean_KiB': statistics.mean(v)}
    summary = {'round': tag, 'kernels': pmc}
    for k, v in pmc.items():
        train = ('patch_kernel' in k and ', 1>' in k) or ('patch_v2_kernel' in k and ', 1, 1, false>' in k)     # MODE_TRAIN (gather input)
        if train and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            fetch = 2 * v['FETCH_SIZE']['median_KiB'] * 1024
            write = v['WRITE_SIZE']['median_KiB'] * 1024
            summary['patch_kernel_hbm_bytes_per_launch'] = fetch + write
            summary['patch_kernel_fetch_bytes_corrected'] = fetch
            summary['patch_kernel_write_bytes'] = write
    json.dump(summary, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
    json.dump(summary, open(os.path.join(out, tag + '_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(summary, indent=1)[:1500])
    # ---- attention network: kernel stats of a train step + matrix-core counters of the attention kernels
    stats = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof_' + tag + 'attn', '**', '*_kernel_stats.csv'), recursive=True))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(os.path.join(out, tag + '_attn_kernel_stats.csv'), 'w', newline='') as f:
            csv.writer(f).writerows(rows[:10])
    files = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', 'pmc_attn', '**', '*_counter_collection.csv'), recursive=True))
    if files:
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        seen = set()
        for r in csv.DictReader(open(files[0])):
            if 'attn_train_kernel' not in r['Kernel_Name']:
                continue
            k = 'attn_train_kernel<..., TRAIN=%s>' % ('true' if 'true>' in r['Kernel_Name'] or ', 1>' in r['Kernel_Name'] else 'false')
            d[k][r['Counter_Name']].append(float(r['Counter_Value']))
            key = (k, r.get('Dispatch_Id'))
            if key not in seen and r.get('End_Timestamp') and r.get('Start_Timestamp'):
                seen.add(key)
                dur[k].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
        res = {'command': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace '
                          '--output-format csv -- python3 tools/attn_bench.py 1024 40 train', 'batch': 1024, 'kernels': {}}
        for k, c in d.items():
            med = {n: statistics.median(v) for n, v in c.items()}
            e = {'median_counters': med, 'launches': len(next(iter(c.values())))}
            if dur[k]:
                e['median_duration_ns_under_pmc'] = statistics.median(dur[k])
            if 'SQ_INSTS_MFMA' in med:
                e['mfma_instructions_per_launch'] = med['SQ_INSTS_MFMA']
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in med and 'GRBM_GUI_ACTIVE' in med and med['GRBM_GUI_ACTIVE'] > 0:
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs can each be MFMA-busy every cycle
                e['mfma_busy_fraction_of_all_simds'] = med['SQ_VALU_MFMA_BUSY_CYCLES'] / (med['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)
            res['kernels'][k] = e
        json.dump(res, open(os.path.join(out, tag + '_attn_mfma.json'), 'w'), indent=1)
        print(json.dumps(res, indent=1)[:1500])




def issue_counts(tag):
    """gpurun_out/pmc_valu{1,4,pan}: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES of every dmf kernel (rocprofv3 --pmc ... on
    bench.py --config 1 / 4 / panms, eager launches) -> profiles/<tag>_issue_counts.json + profiles/issue_counts.json (bench.py
    reads the latter for its `issue_roofline`).  The counters are per launch, summed over all waves of the grid."""
    res = {'round': tag, 'counters': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace', 'configs': {}}
    for name, cfgname in (('pmc_valu1', '1'), ('pmc_valu4', '4'), ('pmc_valupan', 'panms')):
        files = newest(glob.glob(os.path.join(ROOT, 'gpurun_out', name, '**', '*_counter_collection.csv'), recursive=True))
        if not files:
            continue
        d = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(files[0])):
            if 'dmf::' in r['Kernel_Name']:
                d[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        kern = {}
        for k, c in d.items():
            short = k.split('(')[0].replace('void ', '')
            kern[short] = {n: statistics.median(v) for n, v in c.items()}
            kern[short]['launches'] = len(next(iter(c.values())))
        res['configs'][cfgname] = kern
    json.dump(res, open(os.path.join(ROOT, 'profiles', tag + '_issue_counts.json'), 'w'), indent=1)
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'issue_counts.json'), 'w'), indent=1)
    for cfgname, kern in res['configs'].items():
        for k, v in kern.items():
            print(cfgname, k[:90], {n: int(x) for n, x in v.items()})


def configs_md(tag):
    """gpurun_out/<tag>_config_lines.jsonl (tools/collect_r2.sh: a '# command' line, then bench.py's JSON line) -> profiles/<tag>_configs.md"""
    src = os.path.join(ROOT, 'gpurun_out', tag + '_config_lines.jsonl')
    if not os.path.exists(src):
        return
    rows, cmd = [], None
    for line in open(src):
        line = line.strip()
        if line.startswith('#'):
            cmd = line[2:]
        elif line.startswith('{'):
            d = json.loads(line)
            r = d['roofline']
            extra = ''
            if 'loss_scaler' in d:
                extra = '; loss scale %g, %d skipped steps' % (d['loss_scaler']['scale'], d['loss_scaler']['skipped_steps'])
            rows.append('| `%s` | %s | %.3g %s, %.1f us/step; dominant kernel %.2f us, %s fraction %.3f (%s)%s |' % (
                cmd, d['config']['workload'].split(';')[0], d['value'], d['unit'], d['ms_per_step'] * 1e3, r['kernel_ms'] * 1e3,
                'HBM-roof' if r['bound'] == 'hbm' else 'MFMA-peak', r['frac'], d['dtype'].split(',')[0], extra))
    with open(os.path.join(ROOT, 'profiles', tag + '_configs.md'), 'w') as f:
        f.write('# Round-%s bench lines of the BASELINE.json configurations and the batch sweep (1 x MI355X; from the repo root)\n\n' % tag[1:])
        f.write('`value` counts training patches (config 4: stacked stream patches); kernel time = HIP events in `bench.py`.\n\n')
        f.write('| command | workload | result |\n|---|---|---|\n' + '\n'.join(rows) + '\n')


if __name__ == '__main__':
    main()
    configs_md(sys.argv[1] if len(sys.argv) > 1 else 'r1')
    issue_counts(sys.argv[1] if len(sys.argv) > 1 else 'r1')
