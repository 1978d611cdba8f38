import os, sys, subprocess, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for stop in (1,2,3,4,5,0):
    env=dict(os.environ, CTAG_DBG_QUAD_STOP=str(stop))
    out=subprocess.run([sys.executable, os.path.join(ROOT,"bench.py"),"--steps","2","--warmup","1","--frames","1024","--cpu-frames","0"],env=env,capture_output=True,text=True).stdout
    j=json.loads(out.strip().splitlines()[-1])
    print("stop after phase",stop,"quad ms",j["stage_ms_per_step"]["quad"], flush=True)
