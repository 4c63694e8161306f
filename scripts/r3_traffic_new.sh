# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes; kernel-trace only) and instruction counters of the two kernels
# new in round 3: the nearest kernel (8 x 256^3 int16 labels, affine + elastic) and the drawing kernel (134 M draws + sum)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B=$R/tests/native/_build/resample_bench; O=$R/gpurun_out/pmc_new; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O -o nn_$tag --output-format csv -- $B --cases perf --reps 2 --case "labels i16" --path "fast" > $O/nn_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O -o calib_$tag --output-format csv -- $B --cases calib > $O/calib_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_new'
for path in sorted(glob.glob(root+'/*_counter_collection.csv')):
    sums=collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        n=row['Kernel_Name']
        if 'nearest' in n or 'calib' in n:
            sums[(n[:60],row['Counter_Name'])].append(float(row['Counter_Value']))
    for k,v in sums.items():
        big=[x for x in v if x>0.5*max(v)]
        print(os.path.basename(path)[:34], k, round(sum(big)/len(big),1), len(v))
PY
