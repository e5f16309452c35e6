"""Development: what a kernel launch and a stream synchronisation cost on this box (host side) -- context for launch-bound legs such as cfg5."""
import time, torch
x = torch.zeros(64, device="cuda")
torch.cuda.synchronize()
for n in (1000, 1000):
    t = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(n):
        x.add_(1.0); torch.cuda.synchronize()
    t_sync = time.perf_counter() - t
    print("launch %.2f us issued, %.2f us each drained; launch + synchronize %.2f us" % (1e6 * t_issue / n, 1e6 * t_all / n, 1e6 * t_sync / n))
