"""Diagnostic (NOT product code): which hardware queue does ROCm give a new HIP stream?

Round 6 traced the full-suite segfault to hip::Graph::UpdateStreams (libamdhip64 of this image,
torch 2.10.0+rocm7.0): a graph exec with max_streams_ = n creates n parallel streams at
instantiation; at launch it fills its n - 1 side slots from them, SKIPPING every parallel stream
whose queue object equals the launch stream's -- without a bounds check.  Two or more parallel
streams on the launch stream's queue => it reads past the vector => SIGSEGV in hipGraphLaunch.
This probe reads the same field the runtime compares (stream + 0x1a8 -> object -> vtable slot 2),
offsets taken from the disassembly of that build, and prints the queue id of streams as they are
created and destroyed: the assignment policy decides when the collision can happen."""
import ctypes
import sys

import torch

hip = ctypes.CDLL(torch.__file__.rsplit("/", 1)[0] + "/lib/libamdhip64.so")
FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p)


def queue_id(stream_ptr):
    obj = ctypes.c_void_p.from_address(stream_ptr + 0x1a8).value
    vt = ctypes.c_void_p.from_address(obj).value
    fn = ctypes.c_void_p.from_address(vt + 16).value
    return FN(fn)(obj)


def create():
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return s.value


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    ids = {}

    def name(q):
        return ids.setdefault(q, "Q%d" % len(ids))
    live = [create() for _ in range(10)]
    print("10 new streams:        ", [name(queue_id(s)) for s in live])
    for s in live[0:8:4] + live[1:9:4]:       # destroy streams of two queues only
        hip.hipStreamDestroy(ctypes.c_void_p(s))
    print("destroyed 4 (2 queues)")
    more = [create() for _ in range(8)]
    print("8 more streams:        ", [name(queue_id(s)) for s in more])
    ts = [torch.cuda.Stream() for _ in range(6)]
    print("6 torch pool streams:  ", [name(queue_id(t.cuda_stream)) for t in ts])
    print("torch current (default) stream handle:", torch.cuda.current_stream().cuda_stream)


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


def exec_streams(exec_ptr):
    """(max_streams_, [queue names]) of a hipGraphExec (hip::GraphExec layout of this build)."""
    n = ctypes.c_int32.from_address(exec_ptr + 0x48).value
    b = ctypes.c_void_p.from_address(exec_ptr + 0x1b8).value or 0
    e = ctypes.c_void_p.from_address(exec_ptr + 0x1c0).value or 0
    ptrs = [ctypes.c_void_p.from_address(b + 8 * i).value for i in range((e - b) // 8)]
    return n, [queue_id(p) for p in ptrs]


def graph_probe():
    dev = torch.device("cuda", 0)
    a = torch.zeros(1 << 16, device=dev)
    side = [torch.cuda.Stream() for _ in range(3)]
    cap = torch.cuda.Stream()
    for width in (1, 2, 3, 4):
        g = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.stream(cap):
            g.capture_begin(capture_error_mode="thread_local")
            a.add_(1.0)
            for s in side[:width - 1]:
                s.wait_stream(cap)
                with torch.cuda.stream(s):
                    a.mul(2.0)
            for s in side[:width - 1]:
                cap.wait_stream(s)
            a.add_(1.0)
            g.capture_end()
        for attempt in range(3):
            g.instantiate()
            n, qs = exec_streams(g.raw_cuda_graph_exec())
            print(f"width {width}, instantiation {attempt}: max_streams_ {n}, parallel streams on "
                  f"queues {qs}")
            extra = create()        # perturb the balance between instantiations
        g.replay()
    torch.cuda.synchronize()
    print("replayed all; a[0] =", float(a[0]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "graph":
    graph_probe()
