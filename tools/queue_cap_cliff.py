#!/usr/bin/env python3
"""Where the cliff of the queue-cap sweep sits relative to the number of victims (profiles/r06_queue_cap_cliff.txt): python tools/queue_cap_cliff.py 4096 16,32,64,96,128,160,204,4096"""
import ctypes as C, sys, time
sys.path.insert(0,'/root/repo')
from consul_amd import abi
from consul_amd.sim import Sim, preset
import numpy as np
ora = abi.bind(C.CDLL('/root/repo/oracle/_build/libswim_oracle.so'))
n=int(sys.argv[1]); nv=n//20
vict=np.random.default_rng(44).choice(n,size=nv,replace=False).tolist()
for cap in [int(c) for c in sys.argv[2].split(',')]:
    s=Sim(ora, preset(ora, abi.PRESET_LAN, n_nodes=n, seed=11, queue_cap=cap, inbox_cap=2*nv+256, view_cap=nv+64, subject_cap=4))
    s.step_ms(1000); s.kill(0,vict); done=None
    for sec in range(1,500):
        s.step_ms(1000)
        if sec%5==0:
            p,by=s.detection(0)
            if by[2]+by[3]==p: done=sec; break
    st=s.stats()
    print(f"n {n} victims {nv} cap {cap:5d} ({cap/nv:.2f} x victims): full detection at {done} s, drops {st['queue_drops']}, msgs sent per applied {sum(st['msgs_sent'])/max(1,sum(st['msgs_applied'])):.1f}", flush=True)
    s.close()
