#!/usr/bin/env python3
"""Per-stage ALGORITHMIC roofline table of one frame (profiles/<round>_stage_roofline.md).

    python tools/stage_roofline.py --trace <kernel_trace.csv> --detail <bench --detail json>
        [--fetch <counter_collection.csv of a FETCH_SIZE pass>] [--write <... WRITE_SIZE pass>]
        [--dtype f32|bf16] --out profiles/r03_stage_roofline

* durations: rocprofv3 --kernel-trace of `tools/seg_bench.py` (every segment of the frame replayed
  ALONE from its captured graph, the 4 distinct frames in turn): the dispatches are walked in order
  and assigned to the stage they belong to (the launch sequence of sf_lidar_front is fixed);
* algorithmic bytes / flops: `bench.py --detail` (SURVEY 8d formulas, recomputed from the actual
  rulebooks, mean of the same 4 frames);
* counter bytes: two separate rocprofv3 --pmc passes over an eager run of the same frames
  (FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md prescribes, WRITE_SIZE as read; KiB).
"""
import argparse
import collections
import csv
import json
import re

PEAK_HBM = 8.0e12
PEAK_MFMA = {'f32': 157.3e12, 'bf16': 2516.6e12}


def short(name):
    n = re.sub(r'\(anonymous namespace\)::', '', name)
    n = re.sub(r'^void ', '', n)
    m = re.match(r'(\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else n


class Walker:
    """Assigns a dispatch to a stage from its kernel name and the dispatches before it."""

    def __init__(self, conv_rows):
        self.conv_rows = conv_rows      # kernel name -> row labels of its launches, in frame order
        self.seen = collections.Counter()
        self.strided = 0
        self.plan = 0
        self.frames = 0

    def row(self, k):
        base = k.split('<')[0]
        if base == 'fill_many_kernel':
            self.strided, self.plan = 0, 0
            self.frames += 1
            return 'fills'
        if base.startswith('vox_'):
            return 'voxelize'
        if base in ('hash_build_fill_kernel', 'hash_build_kernel'):
            return 'hash'
        if base in ('outset_insert_slots_kernel', 'outset_insert_kernel'):
            self.strided += 1
            return f'rulebook_{self.strided}'
        if base in ('subm27_table_kernel', 'neighbor_table_kernel', 'subm_table_kernel') and self.strided == 0:
            return 'rulebook_0'
        if base.startswith(('outset_', 'nbr_fill', 'scatter27', 'subm27', 'table_fill', 'neighbor_table',
                            'strided_', 'fill_int', 'pairs_to_table')):
            return f'rulebook_{self.strided}'
        if base in ('mask_hist_kernel', 'hist_zero_kernel'):
            if base == 'mask_hist_kernel':
                self.plan += 1
            return f'plan_{max(self.plan, 1)}'
        if base in ('sort_scatter_kernel', 'tile_plan_kernel', 'tile_schedule_kernel'):
            return f'plan_{max(self.plan, 1)}'
        if base in ('dense_zero_kernel', 'dense_mark_kernel', 'dense_write_kernel', 'dense_kernel',
                    'dense_scatter_kernel'):
            return 'dense'
        if k in self.conv_rows:
            # two groups can share an instantiation (bf16: the SubM layers of the last stage and
            # conv_out): the i-th launch of a kernel in a frame belongs to the i-th layer that uses it
            labs = self.conv_rows[k]
            lab = labs[self.seen[k] % len(labs)]
            self.seen[k] += 1
            return lab
        if base == 'cast_bf16_kernel':
            return 'cast'
        return None


def walk(rows, name_key, conv_rows, value):
    """rows in dispatch order -> {row label: [sum, dispatches]}, number of frames seen."""
    w = Walker(conv_rows)
    acc = collections.OrderedDict()
    for r in rows:
        lab = w.row(short(r[name_key]))
        if lab is None:
            continue
        a = acc.setdefault(lab, [0.0, 0])
        a[0] += value(r)
        a[1] += 1
    return acc, w.frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--trace', required=True)
    ap.add_argument('--detail', required=True)
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--dtype', default='f32')
    ap.add_argument('--title', default='')
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    d = json.load(open(args.detail))
    peak_mfma = PEAK_MFMA[args.dtype]

    conv_rows, conv_launch = {}, {}
    for g, v in d['groups'].items():
        conv_launch['conv ' + g] = v['launches']
    for lay in d['layers']:
        g = f"{lay['cin']}x{lay['cout']}" + ('b' if lay['bf16'] else '') + \
            ('' if lay['K'] == 27 else f"k{lay['K']}") + ('' if lay['subm'] else 's')
        conv_rows.setdefault(lay['kernel'], []).append('conv ' + g)

    tr = list(csv.DictReader(open(args.trace)))
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    dur, frames = walk(tr, 'Kernel_Name', conv_rows, lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    if frames == 0:
        raise SystemExit('no front replay (fill_many_kernel) in the trace')

    traffic = {}
    if args.fetch and args.write:
        for path, cname, mul in ((args.fetch, 'FETCH_SIZE', 2048.0), (args.write, 'WRITE_SIZE', 1024.0)):
            rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == cname]
            rows.sort(key=lambda r: int(r['Dispatch_Id']))
            acc, fr = walk(rows, 'Kernel_Name', conv_rows, lambda r: float(r['Counter_Value']) * mul)
            for lab, (s, n) in acc.items():
                per = conv_launch.get(lab)
                # front / dense rows: per frame = sum / frames; conv rows: per launch x launches per frame
                traffic.setdefault(lab, {})[cname] = (s / n * per) if per else s / max(fr, 1)

    st = d['stages']
    rbs = sorted((v['rulebook'], k, v) for k, v in st.items() if 'pairs' in v)
    # the strided rulebook s and the SubM table of its output set share a launch: one row
    rb_rows = collections.OrderedDict()
    si = 0
    for _, name, v in rbs:
        if name.startswith('strided'):
            si += 1
        lab = f'rulebook_{si}'
        e = rb_rows.setdefault(lab, dict(names=[], bytes=0.0, probe=0.0, pairs=0.0))
        e['names'].append(name)
        e['bytes'] += v['bytes']
        e['probe'] += v['probe_bytes']
        e['pairs'] += v['pairs']
    plans = sorted((v['rulebook'], k, v) for k, v in st.items() if k.startswith('plan'))

    table = []   # (label, description, bytes, flops, us per frame, launches per frame)

    def add(lab, desc, by, fl=0.0):
        if lab not in dur:
            return
        s, n = dur[lab]
        per = conv_launch.get(lab)
        us = (s / n * per if per else s / frames) / 1e3
        table.append((lab, desc, by, fl, us, per if per else n / frames))

    v = st['voxelize']
    add('voxelize', f'hard voxelize + mean ({v["points"]:.0f} points -> {v["voxels"]:.0f} voxels)', v['bytes'])
    add('fills', 'table / column prefill of every stage (one launch)', 0.0)
    add('hash', 'coordinate hash of the voxel set', st['voxelize']['voxels'] * 16.0)
    for lab, e in rb_rows.items():
        add(lab, ' + '.join(e['names']) + f' ({e["pairs"]:.0f} pairs)', e['bytes'])
    for i, (_, name, pv) in enumerate(plans):
        add(f'plan_{i + 1}', f'{name}: mask-order plan ({pv["rows"]:.0f} rows; no reference counterpart)', pv['bytes'])
    for g, gv in d['groups'].items():
        add('conv ' + g, f'{gv["launches"]} x ' + ', '.join(gv['kernels']), gv['bytes'], gv['flops'])
    if 'cast' in dur:
        add('cast', 'fp32 -> bf16 copy of the first layer output', 0.0)
    dv = st['dense']
    add('dense', f'dense BEV map ({dv["rows"]:.0f} rows)', dv['bytes'])

    out = [f'# Per-stage algorithmic roofline, one frame{(" -- " + args.title) if args.title else ""}', '',
           'Durations: rocprofv3 `--kernel-trace` of `tools/seg_bench.py` (front / narrow / wide graph of a frame',
           'replayed ALONE, mean of the 4 distinct frames); algorithmic bytes and flops: `bench.py --detail`',
           '(SURVEY 8d formulas on the actual rulebooks, mean of the same frames); floor = max(bytes / 8 TB/s,',
           f'flops / {peak_mfma / 1e12:.1f} TFLOP/s); counter bytes: separate FETCH_SIZE (x2, gfx950) and WRITE_SIZE passes over',
           'an eager run of the same frames.  Hash probes (27 x 8 bytes per row) are NOT in the algorithmic bytes',
           'of a rulebook (SURVEY 8d lists them apart): they are the "probe MB" column.', '',
           '| stage | what | launches / frame | algorithmic MB | GFLOP | µs / frame (alone) | bound | floor µs | fraction of roofline | achieved | counter MB (fetch + write) |',
           '|---|---|---|---|---|---|---|---|---|---|---|']
    tot_us = tot_floor = 0.0
    js = []
    for lab, desc, by, fl, us, n in table:
        t_h, t_m = by / PEAK_HBM * 1e6, fl / peak_mfma * 1e6
        floor = max(t_h, t_m)
        bound = 'mfma' if t_m > t_h else 'hbm'
        ach = (f'{fl / us / 1e6:.1f} TFLOP/s' if bound == 'mfma' else f'{by / us / 1e3:.0f} GB/s') if us > 0 else '-'
        tr_ = traffic.get(lab)
        ctr = f'{tr_.get("FETCH_SIZE", 0) / 1e6:.1f} + {tr_.get("WRITE_SIZE", 0) / 1e6:.1f}' if tr_ else '-'
        probe = ''
        if lab in rb_rows:
            probe = f'; probes {rb_rows[lab]["probe"] / 1e6:.1f} MB'
        frac = f'{floor / us:.1%}' if by or fl else '-'
        out.append(f'| {lab} | {desc}{probe} | {n:.0f} | {by / 1e6:.1f} | {fl / 1e9:.2f} | {us:.1f} | {bound if by or fl else "-"} | '
                   f'{floor:.1f} | {frac} | {ach if by or fl else "-"} | {ctr} |')
        tot_us += us
        tot_floor += floor
        js.append(dict(stage=lab, what=desc, launches=n, algorithmic_bytes=by, flops=fl, us=us, bound=bound,
                       floor_us=floor, frac=floor / us if us else None,
                       counter_fetch_bytes=tr_.get('FETCH_SIZE') if tr_ else None,
                       counter_write_bytes=tr_.get('WRITE_SIZE') if tr_ else None))
    out += ['', f'Sum of the stages alone: {tot_us:.0f} µs per frame; sum of the floors: {tot_floor:.0f} µs '
            f'({tot_floor / tot_us:.1%}).  In the running pipeline the segments of 4 frames overlap on three streams;',
            'the frame rate of `bench.py` is the reciprocal of the wide-layer stream\'s time per frame, not of this sum.']
    open(args.out + '.md', 'w').write('\n'.join(out) + '\n')
    json.dump(js, open(args.out + '.json', 'w'), indent=1)
    print('\n'.join(out))


if __name__ == '__main__':
    main()
