set -e
mkdir -p gpurun_out
rm -rf gpurun_out/soak
python - <<'PY'
import subprocess, sys, time, os, resource
ZMODE = os.environ.get('ZMODE', '').split()      # default: the reference's noise stream continued on the device
t0 = time.time()
p = subprocess.run([sys.executable, 'run.py', 'tc_gan.run.bptt_cwgan', '--', '--datastore', 'gpurun_out/soak', '--iterations', '3000',
                    '--num-models', '32', '--n_bandwidths', '8', '--seqlen', '120', '--skip-steps', '100', '--disc-layers', '[64,64]',
                    '--dataset-provider', 'fixedtime', '--truth_size', '256', *ZMODE, '--disc-precision', 'bf16', '--critic-iters-init', '5', '--quiet',
                    '--disc-param-save-interval', '500'], capture_output=True, text=True)
print('rc', p.returncode, 'wall %.1f s' % (time.time() - t0))
print(p.stderr[-600:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print('child max RSS MB', ru.ru_maxrss / 1024)
import csv
rows = list(csv.DictReader(open('gpurun_out/soak/learning.csv')))
print(len(rows), 'rows; first', rows[0]['Gloss'], 'last', rows[-1]['Gloss'])
import numpy as np
t = np.array([float(r['gen_train_time']) + float(r['gen_forward_time']) + float(r['disc_time']) for r in rows])
print('per-iteration recorded time: first 100 mean %.4f, last 100 mean %.4f' % (t[:100].mean(), t[-100:].mean()))
PY
