import torch
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)*1e3/reps
for mb in (68, 272, 1088):
    y=torch.empty(mb*1000*1000//4, device='cuda'); x=torch.randn_like(y)
    tf=t(lambda: y.fill_(1.0)); tc=t(lambda: y.copy_(x)); tr=t(lambda: x.sum())
    print('%5d MB: fill %6.1f us = %.2f TB/s   copy %6.1f us = %.2f TB/s (r+w)   sum %6.1f us = %.2f TB/s' % (mb, tf, mb/tf, tc, 2*mb/tc, tr, mb/tr))
