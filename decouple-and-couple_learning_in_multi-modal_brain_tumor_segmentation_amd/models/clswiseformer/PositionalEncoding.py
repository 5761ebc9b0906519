"""Fixed positional table kept for checkpoint compatibility (reference PositionalEncoding.py:5-22).

The reference adds ``pe[:x.size(0)]`` with x = [B=1,128,512], i.e. ROW 0 of the table (0 on even, 1 on odd
channels) to every gathered token.  That constant is folded into the gather kernel
(cwf_gather_tokens, pe_odd = 1); this module only owns the [max_len,1,dim] ``pe`` buffer so that the
state_dict keeps its four ``*.pe`` entries."""
import torch
import torch.nn as nn


class ExtendFixedPositionalEncoding(nn.Module):
    def __init__(self, embedding_dim, max_length=512):
        super().__init__()
        pos = torch.arange(0, max_length, dtype=torch.float).unsqueeze(1)
        freq = torch.exp(torch.arange(0, embedding_dim, 2).float() * (-torch.log(torch.tensor(10000.0)) / embedding_dim))
        table = torch.zeros(max_length, embedding_dim)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table.unsqueeze(0).transpose(0, 1).contiguous())

    #: value added to odd channels of every token (cos(0)); even channels get sin(0) = 0
    PE_ODD = 1.0


class LearnedPositionalEncoding(nn.Module):
    """Parameter holder with the reference's shape (PositionalEncoding.py:46-55).  The reference constructs it as
    ``LearnedPositionalEncoding(129, 512)`` (cls_wise_former.py:87-90), i.e. a [1, 512, 129] parameter that cannot be added to the
    [B, 128, 512] token rows: with ``_pe_type="learned"`` (the factory's own default) the reference model constructs but its forward
    raises.  This package mirrors that: construction works (same state_dict keys and shapes), forward raises."""

    def __init__(self, embedding_dim, seq_length):
        super().__init__()
        self.position_embeddings = nn.Parameter(torch.zeros(1, seq_length, embedding_dim))
