set -u
mkdir -p gpurun_out/ta
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 -L > gpurun_out/ta/avail.txt 2>&1
grep -oE "\b(TA|TCP|TD)_[A-Z0-9_]+" gpurun_out/ta/avail.txt | sort -u > gpurun_out/ta/names.txt
wc -l gpurun_out/ta/names.txt
for lib in ${LIBS:-bvh2 bvh4}; do
  B="python3 bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
  PT_HIP_LIB=$PWD/build/ab/$lib.so timeout -s KILL 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ta/$lib -o ta -- $B > gpurun_out/ta/$lib.log 2>&1 || { echo "pass failed $lib"; tail -5 gpurun_out/ta/$lib.log; exit 1; }
  PT_HIP_LIB=$PWD/build/ab/$lib.so timeout -s KILL 200 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d gpurun_out/ta/${lib}_b -o ta -- $B > gpurun_out/ta/${lib}_b.log 2>&1 || { echo "pass b failed $lib"; tail -5 gpurun_out/ta/${lib}_b.log; }
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/ta/bvh*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'pt_trace_items' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print(d, {k: sum(v)/len(v) for k,v in agg.items()})
PY
