"""CPU oracle for the frozen extractors.  TEST INFRASTRUCTURE.

The architecture code is the installed `transformers` package -- the very classes the reference instantiates
(text.py:166-173 AutoModel -> LlamaModel; audio.py:47 Wav2Vec2BertModel; video.py:247 VJEPA2Model) -- built from a
local config with seeded random weights (checkpoints are remote-only: real-checkpoint parity is UNPINNED).  The
reference's post-processing around the model call is restated here with file:line citations.
"""

from __future__ import annotations

import typing as tp

import numpy as np
import torch


def llama_word_states(hf_model: torch.nn.Module, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                      target_words: tp.Sequence[str], pad_id: int) -> list[np.ndarray]:
    """data_utils/features/text.py:235-256 around the HF forward."""
    with torch.no_grad():
        outputs = hf_model(input_ids=input_ids, attention_mask=attention_mask, output_hidden_states=True)  # :236
    hidden_states = torch.stack([layer.cpu() for layer in outputs.hidden_states])  # :240
    out = []
    for i, target_word in enumerate(target_words):  # :243
        hidden_state = hidden_states[:, i]
        n_pads = int((input_ids[i].cpu().numpy() == pad_id).sum())  # :247
        if n_pads:
            hidden_state = hidden_state[:, :-n_pads]  # :250
        word_state = hidden_state[:, -len(target_word):]  # :252
        out.append(word_state.mean(axis=1).cpu().numpy())  # :254-255
    return out
