bash scripts/pmc_traffic.sh > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections,os
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_traffic'
for path in sorted(glob.glob(root+'/*_counter_collection.csv')):
    sums=collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        n=row['Kernel_Name']
        if 'resample' in n or 'plan_bricks' in n or 'calib' in n:
            sums[(n[:70],row['Counter_Name'])].append(float(row['Counter_Value']))
    for k,v in sums.items():
        big=[x for x in v if x>0.5*max(v)]
        print(os.path.basename(path)[:22], k, round(sum(big)/len(big),1), len(v))
PY
