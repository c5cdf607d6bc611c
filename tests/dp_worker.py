"""Worker of tests/test_gpu_dp.py: one rank of a data-parallel ``VAE.fit`` (the product path) under
torch.distributed.  Launched by torch.distributed.run; every rank uses cuda:0 of the one-GPU box and the
collectives go through gloo (host-staged, vae_eeg_xai_amd/dist.py) -- the control flow, sharding, epsilon
indexing and loss weighting are exactly those of an RCCL run.  Rank 0 writes weights + history."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_fit(n, batch, epochs, L, seed, n_val, lr=1e-4):
    from oracle import vae_oracle as vo
    from vae_eeg_xai_amd.models import Encoder, Decoder, VAE, Adam, EarlyStopping
    x = vo.synth_topomaps(n + n_val, 17)
    vae = VAE(Encoder(L), Decoder())
    vae.compile(Adam(lr))
    es = EarlyStopping("val_loss", patience=50)
    hist = vae.fit(x[:n], x[:n], batch, epochs, validation_data=(x[n:], x[n:]), callbacks=[es], verbose=0, seed=seed)
    out = {"h." + k: np.asarray(v, dtype=np.float64) for k, v in hist.history.items()}
    out.update({"w." + k: v for k, v in vae._net.get_weights_dict().items()})
    out["samples_seen"] = np.asarray(vae._net.samples_seen)
    return out


if __name__ == "__main__":
    import torch
    import torch.distributed as dist
    out_path, n, batch, epochs, L, seed, n_val = sys.argv[1], *map(int, sys.argv[2:8])
    lr = float(sys.argv[8])
    torch.cuda.set_device(0)
    dist.init_process_group(os.environ.get("VEEX_DIST_BACKEND", "gloo"))
    res = run_fit(n, batch, epochs, L, seed, n_val, lr)
    if dist.get_rank() == 0:
        np.savez(out_path, **res)
    dist.barrier()
    dist.destroy_process_group()
