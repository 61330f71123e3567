import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import subprocess
for par in ("1","0"):
    env=dict(os.environ, V4L_PAR=par)
    r=subprocess.run([sys.executable,"-m","pytest","tests/test_gpu_parity.py","-q","-s","-p","no:cacheprovider","-k","test_graph_replay_equals_eager and f32"],env=env,capture_output=True,text=True,cwd="/root/repo")
    print("PAR=",par); print("\n".join(l for l in r.stdout.splitlines() if "worst param" in l or "passed" in l or "failed" in l or l.startswith("E  ")))
