import os, sys, subprocess
shapes = [(256,512,64,0),(256,256,64,0),(256,768,64,0),(256,1024,64,0)]
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
    from kq_ab import check
    N,K,M,dq = (int(v) for v in sys.argv[1:5])
    print(N,K,M,dq, check(N,K,M,bool(dq)), flush=True)
else:
    for sh in shapes:
        r = subprocess.run([sys.executable, __file__] + [str(v) for v in sh], capture_output=True, text=True)
        print(sh, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], (r.stderr.strip().splitlines() or [""])[-1][:200], flush=True)
