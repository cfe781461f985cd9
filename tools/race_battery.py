"""The emulation's barrier-race detector (tests/emul/race_rt.h) over text shapes (small tiles) or the six large reference-made golden cases
(full-size tiles, both constructions, p = 0 / 8000).  usage: race_battery.py small | golden"""
import sys, os, ctypes, time, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in ('', 'tests', 'oracle'): sys.path.insert(0, os.path.join(ROOT, _p))
import numpy as np
import caps_sa_amd
from conftest import LARGE_GOLDEN, large_golden, text_bytes
import oracle as O
def load(lib):
    path=os.path.join(os.path.join(ROOT, 'tests', 'emul'), lib)
    E = caps_sa_amd.CapsLib(path, "caps_sa_emul_"); raw = ctypes.CDLL(path)
    raw.caps_sa_emul_races_found.restype = ctypes.c_ulonglong
    return E, raw
which = sys.argv[1]
E, raw = load('libcaps_sa_emul_small_race.so' if which=='small' else 'libcaps_sa_emul_race.so')
tot=0
def run(T, p, name, sa=None, lcp=None, **kw):
    global tot
    for path in ("auto","classic"):
        os.environ["CAPS_SA_PATH"]=path
        t=time.time()
        SA,LCP,st=E.build(T,p=p,**kw)
        if sa is None: sa,lcp=O.build_sa_lcp(T,p=p if p else 0, idx_bits=kw.get('idx_bits',32))[:2]
        ok=np.array_equal(SA,sa) and np.array_equal(LCP,lcp)
        r=raw.caps_sa_emul_races_found(); raw.caps_sa_emul_races_reset(); tot+=r
        print(name,len(T),p,path,'ok' if ok else 'MISMATCH','direct' if st['path_direct'] else 'classic','races',r,'%.1fs'%(time.time()-t),flush=True)
if which=='golden':
    for name in LARGE_GOLDEN:
        T,sa,lcp=large_golden(name)
        for p in (0,8000): run(T,p,name,sa,lcp)
else:
    rs=np.random.RandomState(7)
    DNA=np.frombuffer(b"ACGT",dtype=np.uint8)
    cases=json.load(open(os.path.join(ROOT, 'tests', 'golden', 'cases.json')))['cases']
    for c in cases[:8]:
        T=text_bytes(c['text']) if 'text' in c else None
        if T is not None and len(T)>=32: run(T, c.get('p',0), 'golden:'+c['name'], np.array(c['sa'],dtype=np.uint32), np.array(c['lcp'],dtype=np.uint32))
    n=60000
    run(rs.choice(DNA,size=n),20,'uniform')
    w=rs.dirichlet([0.4]*4); run(rs.choice(DNA,size=n,p=w),13,'skew')
    T=rs.choice(DNA,size=n)
    for _ in range(5):
        ln=int(rs.randint(100,n//8)); a,b=rs.randint(0,n-ln,size=2); T[b:b+ln]=T[a:a+ln]
    run(T,16,'planted')
    T=rs.choice(DNA,size=n); T[1000:9000]=T[1000]; run(T,16,'runs')
    T=np.tile(rs.choice(DNA,size=37),n//37+1)[:n]; run(T,16,'periodic37')
    T=rs.randint(0,200,size=n).astype(np.uint8); run(T,9,'bytes')
    T=rs.choice(DNA,size=n); unit=rs.choice(DNA,size=171)
    T[5000:5000+171*120]=np.tile(unit,120); run(T,16,'tandem')
    run(rs.choice(DNA,size=50001),12,'u64',idx_bits=64)
print('TOTAL races',tot)
