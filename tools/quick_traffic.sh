cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C; timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- $CMD > /tmp/p_$C.log 2>&1
  python - $C <<'PY'
import csv,glob,sys,collections
c=sys.argv[1]
f=glob.glob(f"/tmp/p_{c}/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]==c and "k_encode_blocks" in r["Kernel_Name"]: acc[c].append(float(r["Counter_Value"]))
v=acc[c]; print(c, "KiB per launch k_encode_blocks:", sum(v)/len(v))
PY
done
