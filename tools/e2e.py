"""End-to-end wall time of the drop-in script on a synthetic frappe-sized libfm dataset (202,027 / 57,722 / 28,860 rows
as README.md:37-38): loader, packing into HBM, epochs with their three evaluations - what a user of the reference sees."""
import sys, os, time, tempfile, io, contextlib, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cffm_amd import synth
from cffm_amd import CFFM as host
tmp = tempfile.mkdtemp()
t0 = time.time()
synth.write_libfm(tmp + '/', 'frappe', 5382, 10, 202027, 57722, 28860, dist='zipf')
print('wrote synthetic libfm files in %.1f s' % (time.time() - t0))
os.chdir(tmp)
argv = ['--path', tmp + '/', '--dataset', 'frappe', '--epoch', '3', '--batch_size', '256', '--inner_dims', '32', '--outer_dims', '32',
        '--lr', '0.05', '--num_field', '10', '--activation', 'selu', '--lamda', '0', '--verbose', '1']
t0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    host.main(argv)
print('CFFM.main: %.1f s wall for load + init eval + 3 epochs' % (time.time() - t0))
for line in open('logging.log'):
    if 'Epoch' in line or 'Init_RMSE' in line:
        print(line.rstrip()[:200])
