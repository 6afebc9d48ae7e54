cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lvl in 1 2; do
  export URGYM_SETUP_CACHE=$lvl
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/sc2t/l${lvl}_$c -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sc2t/l${lvl}_$c.log 2>&1
  done
done
python3 - <<'PY'
import glob, pandas as pd, os
R=os.environ['GRAFT_REPO_ROOT']
for lvl in (1,2):
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        f=glob.glob(f'{R}/gpurun_out/sc2t/l{lvl}_{c}/*/*counter_collection.csv')+glob.glob(f'{R}/gpurun_out/sc2t/l{lvl}_{c}/*counter_collection.csv')
        df=pd.read_csv(f[0]); df=df[df.Kernel_Name.str.contains('env_step_fused')]
        print('level',lvl,c,'MB per launch', round(df.Counter_Value.mean()*1024/1e6,1))
PY
rm -rf $R/gpurun_out/sc2t/l*_*/
