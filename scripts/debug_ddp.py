import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dinov2_distillation_amd import ops
    from dinov2_distillation_amd.ddp import init_process_group_from_env, GradBucketReducer
    from dinov2_distillation_amd.trainer import DistillationStepper
    torch.cuda.set_device(0)
    init_process_group_from_env("gloo")
    st = DistillationStepper(device="cuda:0", bucket_bytes=8 << 20)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1234 + rank)).cuda()
    st.reducer.remove()
    st.optimizer.zero_grad(); st.module.training_step(x, 0).backward(); ops.wgrad_join(); torch.cuda.synchronize()
    local = st.optimizer.flat_g.clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered) / world
    class Probe(GradBucketReducer):
        variant = "plain"
        sums = None
        def _on_grad(self, p):
            import traceback
            fr = [f.name for f in traceback.extract_stack()[-4:-1]]
            self.counts.setdefault(id(p), []).append("/".join(fr))
            super()._on_grad(p)
        def _launch(self, bi):
            a, b, _ = self.buckets[bi]
            if self.variant == "devsync":
                torch.cuda.synchronize()
            if self.variant != "nowait":
                self.side.wait_stream(torch.cuda.current_stream()); self.side.wait_stream(ops.wgrad_stream())
                with torch.cuda.stream(self.side):
                    self.sums.append((bi, self.flat_g[a:b].abs().sum()))
            super()._launch(bi)
    for mode in ("plain",):
        ops.ASYNC_WGRAD = True
        st.reducer = Probe(st.optimizer, 8 << 20)
        st.reducer.variant = mode; st.reducer.sums = []; st.reducer.counts = {}
        st.optimizer.zero_grad(); st.module.training_step(x, 0).backward()
        nl = sum(st.reducer.launched)
        st.reducer.finish(); ops.wgrad_join(); torch.cuda.synchronize()
        got = st.optimizer.flat_g
        if rank == 0:
            print(mode, "launched during bwd", nl, "of", len(st.reducer.buckets), "norms got/expect/local", got.norm().item(), expect.norm().item(), local.norm().item())
            for bi, sm in st.reducer.sums:
                a, b, _ = st.reducer.buckets[bi]
                print("   bucket", bi, "abs-sum seen on comm stream before exchange %.4e  local %.4e  final %.4e expect %.4e" % (sm.item(), local[a:b].abs().sum().item(), got[a:b].abs().sum().item(), expect[a:b].abs().sum().item()))
            names = {id(p): n for n, p in st.module.named_parameters()}
            bad = 0
            for p, off, n in st.optimizer.layout:
                c = st.reducer.counts.get(id(p), [])
                if len(c) != 1:
                    print("   NOTIFY x%d %s %s" % (len(c), names[id(p)], c))
            for p, off, n in st.optimizer.layout:
                ex, gt, lc = expect[off:off + n], got[off:off + n], local[off:off + n]
                sc = ex.abs().max().item() + 1e-30
                e = (gt - ex).abs().max().item() / sc
                if e > 1e-4 and bad < 6:
                    bad += 1
                    e_local = (gt - lc).abs().max().item() / sc       # == un-averaged local gradient?
                    e_half = (gt - lc / 2).abs().max().item() / sc    # == local / world without the sum?
                    print("   BAD %-60s bucket %d n=%d err %.1e | vs local %.1e | vs local/2 %.1e" % (names[id(p)][-60:], st.reducer.param_bucket[id(p)], n, e, e_local, e_half))
        st.reducer.remove()
    dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2)
