"""Debug helper: one sequence of the host-library fuzz.  usage: python tests/campaigns/host_fuzz_one.py seed w h cs og"""
import sys, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import pyoracle
import test_host_library as T
T._build()
seed, fw, fh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cs, og = float(sys.argv[4]), float(sys.argv[5])
with tempfile.TemporaryDirectory() as tmp:
    print(T.run_host_operation_sequence(pyoracle, tmp, seed, fw, fh, cs, og))
