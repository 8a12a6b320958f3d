cd $GRAFT_REPO_ROOT
mkdir -p /tmp/hb && cd /tmp/hb
python - <<'PY'
import ctypes as C, os, time, sys, glob
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from basevarc_amd import build as b
exe, hostlib = b.build_host()
H=C.CDLL(hostlib)
H.bvchost_write_synth_batches.restype=C.c_int64
H.bvchost_write_synth_batches.argtypes=[C.c_char_p,C.c_int32,C.c_int32,C.c_int32,C.c_int32,C.c_int32,C.c_uint64,C.c_int32]
n,npos,thread,batch=100000,200,4,500
out='/tmp/hb/o'
for t in range(thread): os.makedirs(f"{out}.tmp.thread.{t}",exist_ok=True)
t0=time.time(); print(H.bvchost_write_synth_batches(out.encode(),n,npos,thread,batch,100,11,2), time.time()-t0)
fs=glob.glob(out+".tmp.thread.*/*")
t0=time.time()
for f in fs:
    with open(f,'rb') as fh: fh.read(16)
print(len(fs),'opens', time.time()-t0)
import subprocess
print(subprocess.run("df /tmp | tail -1; mount | grep -E ' /tmp| / ' | head -3; nproc", shell=True, capture_output=True, text=True).stdout)
PY
cd $GRAFT_REPO_ROOT
g++ -O2 -pthread -o /tmp/t_open tools/_t_open.cpp -ldl
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/basevarc_amd:/opt/rocm/lib /tmp/t_open 0
LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/basevarc_amd:/opt/rocm/lib /tmp/t_open 1
