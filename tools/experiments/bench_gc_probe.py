import gc, time, sys, os
sys.argv=['bench.py','--workload','pcm16_stream','--steps','10','--warmup','3','--cpu-streams','0']
os.environ['AUKIT_BENCH_KEEP_GC']='1'
pauses=[]
t=[0]
def cb(phase, info):
    if phase=='start': t[0]=time.perf_counter()
    else: pauses.append((info['generation'], (time.perf_counter()-t[0])*1e3, info['collected']))
gc.callbacks.append(cb)
import runpy
try:
    runpy.run_path('bench.py', run_name='__main__')
except SystemExit: pass
big=[p for p in pauses if p[1]>1.0]
print('collections', len(pauses), 'over 1 ms:', big[:20], file=sys.stderr)
