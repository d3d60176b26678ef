import torch, statistics
x = torch.zeros(1<<20, device="cuda")
torch.cuda.synchronize()
def pairs(fn, n=200):
    ev=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a,b))
    torch.cuda.synchronize()
    v=[a.elapsed_time(b)*1e3 for a,b in ev]
    return statistics.median(v), min(v), sum(v)/len(v)
print("empty pair us (median,min,mean):", pairs(lambda: None))
print("tiny kernel pair:", pairs(lambda: x[:64].add_(1)))
print("4MB add pair:", pairs(lambda: x.add_(1)))
