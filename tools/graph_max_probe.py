"""Probe (run on the GPU box): is torch's single-stage max over 2048 points per output wrong under
HIP-graph replay, and from which replay on?  Compares graph and eager results with a CPU computation
for several ways of refreshing the input between replays.  Output is quoted in DESIGN.md section 5."""
import torch

dev = torch.device("cuda")


def probe(refresh, staged, shape=(32, 2048, 64), backward=True, replays=4):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(*shape, generator=g).to(dev).requires_grad_(backward)
    w = torch.randn(shape[0], shape[2], generator=g).to(dev)

    def fmax(t):
        if staged:
            B, N, C = t.shape
            return t.view(B, N // 64, 64, C).max(dim=2)[0].max(dim=1)[0]
        return t.max(dim=1)[0]

    def run():
        if backward:
            x.grad = None
        v = fmax(x)
        loss = (v * w).mean()
        if backward:
            loss.backward()
        return v, loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        v_s, loss_s = run()
    grad_s = x.grad if backward else None
    out = []
    for it in range(replays):
        new = torch.randn(*shape, generator=g)
        with torch.no_grad():
            if refresh == "h2d":
                x.copy_(new)
            elif refresh == "h2d+sync":
                x.copy_(new)
                torch.cuda.synchronize()
            elif refresh == "d2d":
                tmp = new.to(dev)
                torch.cuda.synchronize()
                x.copy_(tmp)
        graph.replay()
        torch.cuda.synchronize()
        want_v, want_i = new.max(dim=1)
        ok_graph = torch.equal(v_s.cpu(), want_v)
        with torch.no_grad():
            ok_eager = torch.equal(fmax(x.detach()).cpu(), want_v)
        ok_x = torch.equal(x.detach().cpu(), new)
        ok_grad = None
        if backward:
            gw = torch.zeros_like(new)
            gw.scatter_(1, want_i.unsqueeze(1), (w.cpu() / w.numel()).unsqueeze(1))
            ok_grad = torch.allclose(grad_s.cpu(), gw, rtol=1e-6, atol=1e-9)
        out.append((ok_graph, ok_eager, ok_x, ok_grad))
    return out


for staged in (False, True):
    for backward in (True, False):
        for refresh in ("h2d", "h2d+sync", "d2d"):
            r = probe(refresh, staged, backward=backward)
            print("staged=%-5s backward=%-5s refresh=%-8s (graph==cpu, eager==cpu, x==new, grad==cpu) per replay: %s"
                  % (staged, backward, refresh, r), flush=True)
