"""Diagnostic: lane-mapping variants of the small-E GAE kernel (gae_rtg_chunked_kernel<EW, CW, TC>) at the
headline configuration's rollout (T = 128, E = 4096, fixed-length episodes).  Build: tools/build_variant.sh sweep
-DPPOAF_GAE_SWEEP.  Never shipped."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libppoaf_hip_sweep.so")
from ppo_and_friends_amd import kernels as K

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
NAMES = {0: "EW16 CW8 TC4 (shipped)", 1: "EW32 CW8 TC8", 2: "EW32 CW16 TC4", 3: "EW64 CW8 TC16", 4: "EW64 CW16 TC8",
         5: "EW16 CW16 TC2", 6: "EW32 CW4 TC16", 7: "EW16 CW4 TC8", 8: "EW8 CW8 TC2", 9: "EW32 CW8 TC4", 10: "EW16 CW8 TC2"}


def run(T, E, dense, reps=20):
    r = torch.rand(T, E, device=dev); v = torch.randn(T, E, device=dev)
    adv = torch.empty_like(r); rtg = torch.empty_like(r)
    if dense:
        ek = (torch.rand(T, E, device=dev) < 0.02).to(torch.int8) * 2
        bv = torch.randn(T, E, device=dev); br = torch.randn(T, E, device=dev)
    else:
        ek = None; bv = torch.randn(E, device=dev); br = torch.randn(E, device=dev)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    ref = None
    for var in [int(x) for x in os.environ.get("CHUNKS", "0,1,2,3,4,5,6,7,8,9,10").split(",")]:
        os.environ["PPOAF_GAE_CHUNK"] = str(var)
        for _ in range(3):
            K.gae_rtg_tmajor(r, v, bv, br, ek, adv_out=adv, rtg_out=rtg)
        if ref is None:
            ref = (adv.clone(), rtg.clone())
        else:
            assert torch.equal(adv, ref[0]) and torch.equal(rtg, ref[1]), var
        out = []
        for cold in (False, True):
            us = []
            for _ in range(reps):
                if cold:
                    flush.fill_(1)
                ev = (K.event_create(), K.event_create())
                K.gae_rtg_tmajor(r, v, bv, br, ek, adv_out=adv, rtg_out=rtg, timing_events=ev)
                torch.cuda.synchronize()
                us.append(K.event_elapsed_ms(*ev) * 1e3)
            us.sort()
            out.append((us[0], 0.5 * (us[reps // 2 - 1] + us[reps // 2]), us[-1]))
        (hb, hm, hw), (cb, cm, cw) = out
        print(f"T {T} E {E} dense {int(dense)}  chunk {var:2d} {NAMES[var]:24s} back-to-back best {hb:6.2f} median {hm:6.2f} worst {hw:6.2f} us"
              f"   after a 512 MB flush best {cb:6.2f} median {cm:6.2f} worst {cw:6.2f} us", flush=True)


for T, E, dense in ((128, 4096, False), (128, 4096, True), (128, 16384, False), (200, 1000, True), (1024, 2048, False)):
    run(T, E, dense)
