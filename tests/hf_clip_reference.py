"""Helper run as a SUBPROCESS by test_clip_torch_variant_matches_transformers: a randomly initialised Hugging Face
CLIPTextModel (full vocabulary, random LayerNorm affines) -> its state dict, token ids and last_hidden_state in one
.npz.  Kept out of the test process so torch's bundled HIP runtime never shares a process with libtsd + RCCL."""
import sys

import numpy as np
import torch
import transformers as tr


def main(out_path):
    torch.manual_seed(0)
    cfg = tr.CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                            num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            eos_token_id=49407, bos_token_id=49406, pad_token_id=0)
    m = tr.CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if "layer_norm" in k:
                v.copy_((1.0 if k.endswith("weight") else 0.0) + 0.2 * torch.randn_like(v))
            elif "embedding" in k:
                v.normal_(0, 1.0)
            elif k.endswith("weight"):
                v.normal_(0, 1.0 / np.sqrt(v.shape[1]))
            else:
                v.normal_(0, 0.1)
    tok = np.random.RandomState(3).randint(1, 49405, size=(2, 77))
    tok[1, 40:] = 0  # a padded prompt
    with torch.no_grad():
        ref = m(input_ids=torch.from_numpy(tok)).last_hidden_state.numpy()
    arrays = {"state/" + k: v.numpy() for k, v in m.state_dict().items() if v.dtype.is_floating_point}
    np.savez(out_path, tokens=tok, reference=ref, **arrays)


if __name__ == "__main__":
    main(sys.argv[1])
