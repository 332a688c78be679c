import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build()
maps = open('/proc/self/maps').read()
print('after build:', sorted(set(l.split()[-1] for l in maps.splitlines() if 'amdhip64' in l or 'libdl3p' in l or 'libhsa-runtime' in l)))
import torch
print('torch cuda', torch.cuda.is_available())
try:
    g.smoke()
except Exception as e:
    print('SMOKE FAILED', repr(e)[:300])
maps = open('/proc/self/maps').read()
print('after smoke:', sorted(set(l.split()[-1] for l in maps.splitlines() if 'amdhip64' in l or 'libdl3p' in l or 'libhsa-runtime' in l)))
