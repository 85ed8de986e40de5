#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/fetchcal under rocprofv3 (one PMC pass), per kernel, against the bytes each kernel reads
REPO=$(pwd)
OUT=$REPO/gpurun_out/fetchcal
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- $REPO/tools/fetchcal > $OUT/fetchcal.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
line = [l for l in open(os.path.join(out, 'fetchcal.log')) if l.startswith('bytes_read')][-1].split()
read = {line[i]: float(line[i + 1]) for i in range(1, len(line), 2)}
for f in glob.glob(os.path.join(out, 'pmc/**/*counter_collection.csv'), recursive=True):
	for r in csv.DictReader(open(f)):
		if r.get('Counter_Name') != 'FETCH_SIZE':
			continue
		name = re.sub(r'\(.*', '', r['Kernel_Name'])
		name = re.sub(r'^void ', '', name)
		key = next((k for k in read if k == name or name.startswith(k)), None)
		if key is None:
			continue
		kib = float(r['Counter_Value'])
		print('%-22s read %.3f GB  FETCH_SIZE %.3f GB as reported  ->  factor %.3f' % (name, read[key] / 1e9, kib * 1024 / 1e9, read[key] / (kib * 1024)))
PY
