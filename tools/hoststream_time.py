import sys, os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import numpy as np, time, os, threading
import physicsbasedbayesianinference_amd._hoststream as hs
n=128*65536
for rep in range(3):
    np.random.seed(5); t=time.perf_counter(); z=np.random.standard_normal(n); t1=time.perf_counter()-t
    np.random.seed(5); t=time.perf_counter(); z2=hs.standard_normal(n); t2=time.perf_counter()-t
    print("main thread: numpy %.1f ms  fast %.1f ms  equal %s" % (t1*1e3,t2*1e3,np.array_equal(z,z2)), flush=True)
def work():
    np.random.seed(5); t=time.perf_counter(); z2=hs.standard_normal(n); print("worker thread fast %.1f ms" % ((time.perf_counter()-t)*1e3), flush=True)
for rep in range(2):
    th=threading.Thread(target=work); th.start(); th.join()
print("cpus", len(os.sched_getaffinity(0)))
