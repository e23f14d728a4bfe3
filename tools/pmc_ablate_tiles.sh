cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in abl2 abl1; do
  PIXELSPLAT_HIP_LIB=$R/pixelsplat_amd/libps_$v.so timeout 25 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc_$v -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes --launch eager > /dev/null 2>&1
  python - "$v" <<'P'
import csv,glob,sys,collections
v=sys.argv[1]
f=glob.glob(f'/tmp/pmc_{v}/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for row in csv.DictReader(open(f[0])):
    k=row['Kernel_Name']
    if 'tiles_' not in k: continue
    acc[k[:40]][row['Counter_Name']]+=float(row['Counter_Value'])
    if row['Counter_Name']=='SQ_INSTS_VALU': n[k[:40]]+=1
for k in acc: print(v, k, {c: round(x/n[k]/1e6,1) for c,x in acc[k].items()}, 'launches', n[k])
P
done
