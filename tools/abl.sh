#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" gpurun_abl_*.so; do
  if [ -z "$f" ]; then echo -n "full: "; python bench.py --steps 10 --warmup 2 --no-cpu --no-fri 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" ;
  else echo -n "$f: "; BFS_LIB_PATH=$GRAFT_REPO_ROOT/$f python - <<'PY'
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer
lib=_lib.load()
n=1<<24; cols=8
d_in=DeviceBuffer(n*cols); d_out=DeviceBuffer(n*cols)
root=lib.bfs_gl_primitive_root(24)
for _ in range(2): _lib.check(lib.bfs_gl_ntt(d_in.ptr,n,n,d_out.ptr,n,24,cols,root,1,1,0))
e0,e1=ctypes.c_void_p(),ctypes.c_void_p(); lib.bfs_event_create(ctypes.byref(e0)); lib.bfs_event_create(ctypes.byref(e1))
lib.bfs_event_record(e0,0)
for _ in range(10): _lib.check(lib.bfs_gl_ntt(d_in.ptr,n,n,d_out.ptr,n,24,cols,root,1,1,0))
lib.bfs_event_record(e1,0); ms=ctypes.c_float(); lib.bfs_event_elapsed_ms(e0,e1,ctypes.byref(ms)); print(ms.value/10)
PY
  fi
done
