#!/usr/bin/env python3
"""Static census of tiles_backward_kernel<false>'s gfx950 ISA: registers / LDS / scratch, VALU instructions per quadrant
evaluation block (both forms), per reduction block and per finalisation block -- the numbers DESIGN.md 4a quotes.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ipixelsplat_amd/csrc --cuda-device-only -S \
          pixelsplat_amd/csrc/raster_tiles.hip -o /tmp/tiles.s
    python tools/backward_isa_census.py /tmp/tiles.s > profiles/r6_backward_isa_census.txt
"""
import re
import sys

txt = open(sys.argv[1]).read()
m = re.search(r'^(_ZN2ps21tiles_backward_kernelILb0E\w*):', txt, re.M)
start = m.end()
body = txt[start:txt.index('.Lfunc_end', start)].split('\n')
print("tiles_backward_kernel<false>: static census of the shipped kernel's gfx950 ISA  (tools/backward_isa_census.py)")
print("=" * 110)
print("'evaluation block' = a basic block that contains the v_exp_f32 of one quadrant evaluation (2 forms x 2 entries per trip")
print("x 4 quadrants = 16 of them; the first quadrant of an entry also initialises the entry's accumulators).")
meta = re.search(r'\.amdhsa_kernel _ZN2ps21tiles_backward_kernelILb0E.*?\.end_amdhsa_kernel', txt, re.S).group(0)
for key in ('next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size'):
    val = re.search(r'.amdhsa_%s (\d+)' % key, meta).group(1)
    print(f"  {key:28s} {val}")
blocks, cur = [], []
for line in body:
    s = line.strip()
    if s.startswith(('.LBB', 's_cbranch', 's_branch', '; %bb')):
        if cur:
            blocks.append(cur)
            cur = []
        continue
    cur.append(s)
if cur:
    blocks.append(cur)
print("\nevaluation blocks ('short' = no v_min_f32 = the form for plain entries inside every pixel's walk):")
for b in blocks:
    if any(x.startswith('v_exp_f32') for x in b):
        v = [x for x in b if x.startswith('v_')]
        form = 'short' if not any('v_min_f32' in x for x in b) else 'full'
        print(f"  {form:5s}  VALU {len(v):3d}   packed {sum(x.startswith('v_pk_') for x in v):2d}   v_mov {sum(x.startswith('v_mov') for x in v)}")
print("\nreduction blocks (lane swaps + packed adds + the stage writes), per contributing entry:")
for b in blocks:
    if any('v_permlane32_swap' in x for x in b):
        v = [x for x in b if x.startswith('v_')]
        print(f"  VALU {len(v):3d} ({sum('permlane' in x for x in v)} swaps, {sum(x.startswith('v_pk_add') for x in v)} packed adds)"
              f"   LDS {sum(x.startswith('ds_') for x in b)}")
print("\nfinalisation blocks (eight lanes per entry, per batch of 8 entries; one per form), up to the slot / atomic branch:")
for b in blocks:
    if any('row_half_mirror' in x for x in b):
        print(f"  VALU {sum(x.startswith('v_') for x in b):3d}   LDS reads {sum(x.startswith('ds_') for x in b)}")
tot = [x for x in body if re.match(r'\s+v_', x)]
print(f"\nwhole kernel: {len(tot)} VALU instructions static, {sum(x.strip().startswith('ds_') for x in body)} LDS, "
      f"{sum(x.strip().startswith('global_') for x in body)} global, {sum('scratch_' in x for x in body)} scratch")
