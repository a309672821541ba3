"""Where do the gather kernels stand once the working set no longer fits the 256 MiB Infinity Cache?
K2 / K3 / K4 through the C ABI at (nu, F, members) points, COMPULSORY bytes (every input and output byte
once: x read, out written, indices/weights, W) per launch over the launch time, against 8 TB/s.
    python tools/hbm_regime.py [nu:F:M ...]        default: a sweep around c3 x 4 members and nu=300"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gwen_amd
from gwen_amd import _lib
from gwen_amd.graph import _ptr, _stream


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best


def main():
    pts = sys.argv[1:] or ["100:256:1", "100:256:2", "100:256:4", "100:256:8", "100:64:4", "100:64:16",
                           "300:64:1", "300:256:1"]
    which = os.environ.get("KB_WHICH", "k2,k3,k4,k8").split(",")
    CONTRACT = _lib.CONTRACT_NAMES[os.environ.get("KB_CONTRACT", "3xbf16")]            # K3 / K4 / K8 contraction
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _stream(dev)
    meshes = {}
    for p in pts:
        nu, F, M = (int(v) for v in p.split(":"))
        if nu not in meshes:
            mesh = gwen_amd.geodesic_mesh(nu, reorder=os.environ.get("KB_REORDER", "hilbert"))
            g = gwen_amd.prepare_graph(torch.from_numpy(mesh.edge_index).to(dev), mesh.num_nodes)
            meshes[nu] = (mesh, g, g.grouped())
        mesh, g, (gr, gc, gv) = meshes[nu]
        N, E = mesh.num_nodes, mesh.num_edges
        h = torch.randn(M, N, F, device=dev)
        out = torch.empty(M, N, F, device=dev)
        b = torch.randn(F, device=dev)
        w = torch.randn(F, F, device=dev) / F ** 0.5
        comp = 8 * M * N * F + 8 * (E + N) + 4 * N          # x once, out once, col+val, rowptr
        ws_mb = 2 * 4 * M * N * F / 2 ** 20
        print(f"nu={nu} N={N} E={E} F={F} members={M}: in+out {ws_mb:.0f} MiB, compulsory {comp/1e6:.1f} MB "
              f"(floor {comp/8e6:.1f} us at 8 TB/s)", flush=True)
        t = timed(lambda: out.copy_(h))
        print(f"   copy         {t:8.1f} us  {8*M*N*F/t/1e6:5.2f} TB/s (torch copy_, the streaming rate of this box)", flush=True)
        if "k2" in which:
            t = timed(lambda: L.gwen_gcn_propagate_f32(_ptr(g.rowptr), _ptr(g.col), _ptr(g.val), _ptr(h), _ptr(b),
                                                       _ptr(out), N, F, F, F, M, N * F, N * F, 1, st))
            print(f"   K2 propagate {t:8.1f} us  {comp/t/1e6:5.2f} TB/s compulsory = {comp/t/8e6:.3f} of HBM peak"
                  f"   {M*E/t/1e3:6.2f} Gedge/s", flush=True)
        if "k3" in which:
            nws = int(L.gwen_gcn_linear_workspace_floats(M * N, F, F))
            wsb = torch.empty(max(nws, 1), device=dev)
            t = timed(lambda: L.gwen_gcn_linear_f32(_ptr(h), _ptr(w), _ptr(b), _ptr(out), M * N, F, F, F, F, 1, CONTRACT,
                                                    _ptr(wsb), nws, st))
            c3 = 8 * M * N * F + 4 * F * F
            print(f"   K3 linear    {t:8.1f} us  {c3/t/1e6:5.2f} TB/s compulsory = {c3/t/8e6:.3f}   "
                  f"{2*M*N*F*F/t/1e6:6.1f} TFLOP/s", flush=True)
        if "k4" in which and L.gwen_gcn_layer_supported(F, F):
            t = timed(lambda: L.gwen_gcn_layer_f32(_ptr(gr), _ptr(gc), _ptr(gv), _ptr(h), _ptr(w), _ptr(b), _ptr(out),
                                                   N, F, F, F, F, M, N * F, N * F, 1, CONTRACT, st))
            c4 = comp + 4 * F * F
            print(f"   K4 layer     {t:8.1f} us  {c4/t/1e6:5.2f} TB/s compulsory = {c4/t/8e6:.3f} of HBM peak"
                  f"   {M*E/t/1e3:6.2f} Gedge/s", flush=True)
        if "k8" in which and L.gwen_gcn_wide_supported(F, F) and g.tiles() is not None:
            tr, tl, tv, umax = g.tiles()
            t = timed(lambda: L.gwen_gcn_wide_layer_f32(_ptr(tr), _ptr(tl), _ptr(tv), _ptr(h), _ptr(w), _ptr(b), _ptr(out),
                                                        N, N, F, F, F, M, N * F, N * F, 1, umax, CONTRACT, st))
            c4 = comp + 4 * F * F
            print(f"   K8 wide      {t:8.1f} us  {c4/t/1e6:5.2f} TB/s compulsory = {c4/t/8e6:.3f} of HBM peak"
                  f"   {M*E/t/1e3:6.2f} Gedge/s", flush=True)
        del h, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
