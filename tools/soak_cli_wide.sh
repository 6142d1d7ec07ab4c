# As soak_cli.sh with a WIDE critic (3 x 512: the layer-by-layer path, whose update forks onto three streams):
# 600 iterations through the reference CLI with every recorder on; checks that the run ends, losses stay finite, the
# process does not grow and the recorded time per iteration does not drift.
set -e
mkdir -p gpurun_out
rm -rf gpurun_out/soak_wide
python - <<'PY'
import subprocess, sys, time, os, resource
ZMODE = os.environ.get('ZMODE', '').split()      # default: the reference's noise stream continued on the device
t0 = time.time()
p = subprocess.run([sys.executable, 'run.py', 'tc_gan.run.bptt_cwgan', '--', '--datastore', 'gpurun_out/soak_wide', '--iterations', '600',
                    '--num-models', '256', '--n_bandwidths', '8', '--seqlen', '120', '--skip-steps', '100', '--disc-layers', '[512,512,512]',
                    '--dataset-provider', 'fixedtime', '--truth_size', '512', *ZMODE, '--disc-precision', 'bf16', '--critic-iters-init', '5', '--quiet',
                    '--disc-param-save-interval', '200'], capture_output=True, text=True)
print('rc', p.returncode, 'wall %.1f s' % (time.time() - t0))
print(p.stderr[-600:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print('child max RSS MB', ru.ru_maxrss / 1024)
import csv
import numpy as np
rows = list(csv.DictReader(open('gpurun_out/soak_wide/learning.csv')))
g = np.array([float(r['Gloss']) for r in rows])
print(len(rows), 'rows; first', g[0], 'last', g[-1], 'all finite', bool(np.isfinite(g).all()))
t = np.array([float(r['gen_train_time']) + float(r['gen_forward_time']) + float(r['disc_time']) for r in rows])
print('per-iteration recorded time: first 100 mean %.4f, last 100 mean %.4f' % (t[:100].mean(), t[-100:].mean()))
assert p.returncode == 0 and np.isfinite(g).all()
PY
