import os, sys, socket, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, port, q, sync):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=False, sync_bn=sync, gradient_clipping=0.0)
    im, mk = unet_amd.ellipse_batch(4, 64, seed=21)
    out = st.step(im[rank*2:rank*2+2].to(dev), mk[rank*2:rank*2+2].to(dev))
    torch.cuda.synchronize()
    q.put((rank, st.optimizer.flat_g.cpu().numpy(), float(out["loss"].detach()), float(out["grad_norm"])))
    dist.destroy_process_group()

if __name__ == "__main__":
    import unet_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = unet_amd.UNet_T(1, 1, bilinear=True).to(dev)
    st = unet_amd.TrainStepper(model, lr=1e-4, amp=False, gradient_clipping=0.0)
    im, mk = unet_amd.ellipse_batch(4, 64, seed=21)
    out = st.step(im.to(dev), mk.to(dev))
    torch.cuda.synchronize()
    ref = st.optimizer.flat_g.cpu()
    names = [n for n, _ in model.named_parameters()][::-1]
    slices = st.optimizer.slices
    print("single: loss", float(out["loss"].detach()), "gnorm", float(out["grad_norm"]))
    for sync in (True, False):
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [ctx.Process(target=worker, args=(r, 2, port, q, sync)) for r in range(2)]
        [p.start() for p in ps]
        res = [q.get(timeout=300) for _ in ps]
        [p.join() for p in ps]
        g = torch.from_numpy(res[0][1])
        print(f"sync_bn={sync}: loss {res[0][2]:.5f} gnorm {res[0][3]:.4f}; total rel L2 {float((g-ref).norm()/ref.norm()):.3e}; ratio of norms {float(g.norm()/ref.norm()):.4f}")
        for nme, (o, n) in list(zip(names, slices))[:6] + list(zip(names, slices))[-6:]:
            a, b = g[o:o+n], ref[o:o+n]
            print(f"   {nme:40s} rel {float((a-b).norm()/b.norm().clamp_min(1e-20)):.3e}  norm ratio {float(a.norm()/b.norm().clamp_min(1e-20)):.3f}")
