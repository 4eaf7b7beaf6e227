import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "seal-3d_amd"))
import s3d_hip
s3d_hip.GridBackend.set_backward_path(int(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
import runpy
runpy.run_path(os.path.join(REPO, "bench.py"), run_name="__main__")
