"""CPU emulation: how far can e4m3 Linear GEMMs carry a random-weight Llama stack?  (BASELINE config 5 / VERDICT r1 item 6.)

Every Linear of a 28-layer Llama (real depth, real head geometry; width reduced to keep fp32 CPU time in seconds) is replaced by
a fake-quantised product: inputs and weights are rounded to OCP e4m3 with
  (a) one scale per tensor (what csrc/gemm_fp8.hip + tribe_quantize_fp8_fwd do today), or
  (b) MX block scales: one power-of-two (E8M0) scale per 32 consecutive k of every row, on BOTH operands -- the block-scaled
      form v_mfma_scale_f32_16x16x128_f8f6f4 supports natively,
and the product is accumulated in f32.  Reported: relative L2 error of the hidden states (mean over positions) against the
unquantised model at layers 1 / 7 / 14 / 21 / 28.  Run:  python scripts/fp8_depth_emulation.py [hidden]"""
import sys

import torch
from transformers import LlamaConfig, LlamaModel

hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
cfg = LlamaConfig(vocab_size=400, hidden_size=hidden, intermediate_size=hidden * 8 // 3 // 128 * 128, num_hidden_layers=28,
                  num_attention_heads=hidden // 128, num_key_value_heads=max(1, hidden // 512), head_dim=128, rms_norm_eps=1e-5,
                  max_position_embeddings=4096, tie_word_embeddings=True)
model = LlamaModel(cfg).eval()
ids = torch.randint(8, 400, (2, 128))
F8 = torch.float8_e4m3fn


def q_tensor(t):
    s = t.abs().amax().clamp_min(1e-30) / 448.0
    return (t / s).clamp(-448, 448).to(F8).float() * s


def q_mx(t):
    *lead, k = t.shape
    b = t.reshape(*lead, k // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(2.0**-120)
    s = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))          # E8M0: power of two, no element above 448 after scaling
    return ((b / s).clamp(-448, 448).to(F8).float() * s).reshape(t.shape)


class QLinear(torch.nn.Module):
    def __init__(self, lin, q):
        super().__init__()
        self.w, self.q = q(lin.weight.detach()), q

    def forward(self, x):
        return self.q(x) @ self.w.t()


def run(q):
    m = LlamaModel(cfg).eval()
    m.load_state_dict(model.state_dict())
    if q is not None:
        for layer in m.layers:
            for parent, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
                for n in names:
                    setattr(parent, n, QLinear(getattr(parent, n), q))
    with torch.no_grad():
        return [h.mean(dim=1) for h in m(input_ids=ids, output_hidden_states=True).hidden_states]


ref = run(None)
for name, q in (("per-tensor e4m3", q_tensor), ("MX block-32 e4m3 (both operands)", q_mx)):
    got = run(q)
    errs = {L: float((got[L] - ref[L]).norm() / ref[L].norm()) for L in (1, 7, 14, 21, 28)}
    print(f"hidden {hidden}, {name:34s} relative L2 of pooled states at layers 1/7/14/21/28:", " ".join(f"{e:.3f}" for e in errs.values()))
bf = run(lambda t: t.bfloat16().float())
errs = {L: float((bf[L] - ref[L]).norm() / ref[L].norm()) for L in (1, 7, 14, 21, 28)}
print(f"hidden {hidden}, {'bf16 operands (for scale)':34s} relative L2 of pooled states at layers 1/7/14/21/28:", " ".join(f"{e:.4f}" for e in errs.values()))
