#!/bin/bash
# Wall time of the reference CLI at the paper's shape (scripts/fig4/gan/run.json), run on the GPU box via gpurun:
# 150 generator steps with every recorder on; compare with `bench.py --workload c3paper` (loop only).
# ZMODE: empty (default) = the reference's noise stream continued on the device; '--z-device-seed 5' = Philox; '--z-host-draw'
set -e
ZMODE=${ZMODE:-}
out=gpurun_out/cli_run
rm -rf $out
echo '{"num_sites": 101, "tau_E": 2, "ssn_type": "deg-heteroin", "V0": 0.1}' > gpurun_out/paper_cfg.json
./run tc_gan.run.bptt_cwgan -- --datastore $out --load-config gpurun_out/paper_cfg.json --iterations 150 --num-models 128 \
  --n_bandwidths 8 --seqlen 240 --skip-steps 200 --disc-layers '[128,128,128,128]' --disc-normalization layer \
  --dataset-provider fixedtime --truth_size 512 $ZMODE --disc-precision bf16 --critic-iters-init 5 --quiet \
  --gen-update-name rmsprop --disc-update-name rmsprop --gen-learning-rate 1e-4 --disc-learning-rate 0.02 > gpurun_out/cli_run.log 2>&1
tail -2 gpurun_out/cli_run.log
python3 - <<'PY'
import csv, datetime, re
rows = list(csv.reader(open('gpurun_out/cli_run/learning.csv')))
print(rows[0]); print(rows[-1]); print(len(rows) - 1, 'rows')
ls = [l for l in open('gpurun_out/cli_run.log') if re.match(r'\d{4}-\d\d-\d\d', l)]
t = lambda l: datetime.datetime.strptime(l[:23], '%Y-%m-%d %H:%M:%S,%f')
a = [l for l in ls if 'start iterations' in l][0]
b = [l for l in ls if 'maximum iterations' in l][0]
dt = (t(b) - t(a)).total_seconds()
print('loop %.3f s -> %.2f ms per generator step' % (dt, dt / 150 * 1e3))
PY
