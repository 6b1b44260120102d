import time, sys
sys.path.insert(0,'.')
from helib_amd import host
t=time.time(); s=host.Session("bgv",32003,2,1,5800,8,seed=5); print('setup',time.time()-t, s.L_ctxt,s.K,s.D, flush=True)
from helib_amd import capi as hx
for lvl in (1,2):
    t=time.time(); s.multiply(lvl,1,True); 
    import ctypes
    t1=time.time()-t
    t=time.time(); s.multiply(lvl,4,True); s.ctxt_info(lvl); dt=time.time()-t
    print('level',lvl,'first',t1,'4 multiplies of batch 8:',dt, 'mult/s', 32/dt, flush=True)
    print('verify', s.verify(lvl, elements=[0]), flush=True)
