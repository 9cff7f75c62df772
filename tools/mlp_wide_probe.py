"""Device time of the wide-MLP launches (csrc/mlp_wide.hip) at SAC's shapes: twin critics
(393 -> 256 -> 256 -> 1, [obs 376 | action 17]) and the actor (376 -> 256 -> 256 -> 34) at batch
256; every case captured 20x into a HIP graph and timed with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    from agents_amd.specs import tensor_spec
    from agents_amd.utils import graph

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    B = int(os.environ.get("B", "256"))

    def net(d0, widths, seed):
        n = sequential.Sequential([L.Dense(w, activation=a) for w, a in widths], seed=seed)
        n.create_variables(tensor_spec.TensorSpec((d0,), torch.float32, "x"), device=dev)
        return n

    crit = [net(393, ((256, "relu"), (256, "relu"), (1, None)), 1 + i) for i in range(2)]
    actor = net(376, ((256, "relu"), (256, "relu"), (34, None)), 7)
    obs = torch.randn(B, 376, device=dev)
    act = torch.randn(B, 17, device=dev)
    dq = [torch.randn(B, 1, device=dev) for _ in range(2)]
    dz = torch.randn(B, 34, device=dev)
    dxs = [torch.zeros(B, 393, device=dev) for _ in range(2)]

    def timeit(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        c = graph._Captured()
        c.capture(lambda: [fn() for _ in range(reps)] and None)
        best = 1e9
        for _ in range(5):
            c.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            c.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
        return best * 1e3

    f2 = lambda: sequential.forward_wide(crit, [obs, obs], slot="p", need_grad=True, x2s=[act, act])
    f1 = lambda: sequential.forward_wide(crit[:1], [obs], slot="p", need_grad=True, x2s=[act])
    fa = lambda: sequential.forward_wide([actor], [obs], slot="p", need_grad=True)
    print(f"B={B}")
    print(f"critic pair forward            : {timeit(f2):6.1f} us")
    print(f"one critic forward             : {timeit(f1):6.1f} us")
    print(f"actor forward                  : {timeit(fa):6.1f} us")
    f2()
    fa()
    print(f"critic pair backward (chain+dW): "
          f"{timeit(lambda: sequential.backward_wide(crit, dq, slot='p')):6.1f} us")
    print(f"critic pair chain only, d/d action: "
          f"{timeit(lambda: sequential.backward_wide(crit, dq, slot='p', param_grads=False, input_grads=dxs, input_grad_cols=(376, 393))):6.1f} us")
    print(f"actor backward (chain+dW)      : "
          f"{timeit(lambda: sequential.backward_wide([actor], [dz], slot='p')):6.1f} us")


    # in-kernel timeline of the gradient chain (eager launches: captured graphs hold the old
    # kernel arguments)
    from agents_amd import _lib
    lib = _lib.load()
    for name, fn, n_wg in (
            ("critic pair forward (0 start, 1 input staged, per layer: streamed / summed / out)",
             f2, 2 * ((B + 3) // 4)),
            ("critic pair chain, d/d action", lambda: sequential.backward_wide(
                crit, dq, slot="p", param_grads=False, input_grads=dxs,
                input_grad_cols=(376, 393)), 2 * ((B + 3) // 4)),
            ("actor chain", lambda: sequential.backward_wide(
                [actor], [dz], slot="p", param_grads=False), (B + 3) // 4)):
        buf = torch.zeros((n_wg, 16), dtype=torch.int64, device=dev)
        lib.aa_mlp_wide_debug_stamps(buf.data_ptr())
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        lib.aa_mlp_wide_debug_stamps(None)
        t = buf.cpu().numpy().astype("float64")
        t = t[t[:, 0] > 0]                    # (workgroups that ran: the MFMA form has fewer)
        rel = (t - t[:, :1]) * 0.01           # us since the workgroup's first stamp
        span = (t[:, :11].max() - t[:, 0].min()) * 0.01
        print(f"{name}: stamps (us, median over workgroups) "
              + " ".join(f"{v:5.1f}" for v in __import__("numpy").median(rel[:, :11], axis=0)
                         if v >= 0)
              + f" | first start -> last end {span:5.1f} us | second layer from the top: rows "
              f"requested {__import__('numpy').median(rel[:, 10]):5.1f}, dz computed "
              f"{__import__('numpy').median(rel[:, 11]):5.1f} (negative: not stamped by this kernel)")


if __name__ == "__main__":
    main()
