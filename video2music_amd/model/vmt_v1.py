"""``VideoMusicTransformer_V1`` (reference ``model/video_music_transformer.py:22-314``): the V2 machinery with other layer plans.
Importable from ``video2music_amd.model.video_music_transformer`` like in the reference; split out for size."""
import torch.nn as nn

from ..utilities.constants import CHORD_ATTR_SIZE, CHORD_ROOT_SIZE, CHORD_SIZE, SCENE_OFFSET_MAX
from .vmt_v2 import VideoMusicTransformer_V2, _TransformerParamsV2


class VideoMusicTransformer_V1(VideoMusicTransformer_V2):
    """Reference ``VideoMusicTransformer_V1`` (model/video_music_transformer.py:22-314), eval mode.  Learned positional
    tables on both streams (:63-65,198-206); every layer's feed-forward a 6-expert top-2 mixture -- ``MoELayer`` for
    '1.0', '1.1', '1.3.4', else ``SharedMoELayer`` -- over ``GLUExpert(d, d_ff)`` ('1.1', '1.3') or
    ``Linear(d, 2d) -> SiLU -> Linear(2d, d)`` experts (:77-85); '1.3.3' / '1.3.4' put three plain GLU layers first
    (:108-125); RoPE inside the attentions when ``version_name in '1.2.3'`` -- the reference's substring test (:86), so
    '1.2' gets it too; ``rms_norm=True`` swaps every LayerNorm for RMSNorm (:69-72).  ``nn.MultiheadAttention`` and
    ``CustomMultiheadAttention`` carry the same parameter names and compute the same attention without RoPE, so one
    code path serves both.  forward / generate / the KV-cached decode step are inherited from the V2 class.
    """

    def __init__(self, version_name="1.1", n_layers=6, num_heads=8, d_model=512, dim_feedforward=1024, dropout=0.1,
                 max_sequence_midi=2048, max_sequence_video=300, max_sequence_chord=300, total_vf_dim=0, rms_norm=False,
                 scene_embed=False, chord_embed=False, dropTokenRate=0.0):
        nn.Module.__init__(self)
        from .custom_transformer import RMSNorm
        from .moe import GLUExpert, MoELayer, SharedMoELayer, SiLUExpert
        shallow = version_name in ("1.3.3", "1.3.4")
        self.nlayers, self.nhead, self.d_model, self.d_ff, self.dropout = n_layers, num_heads, d_model, dim_feedforward, dropout
        self.max_seq_midi, self.max_seq_video, self.max_seq_chord = max_sequence_midi, max_sequence_video, max_sequence_chord
        self.scene_embed, self.chord_embed, self.dropTokenRate, self.version_name = scene_embed, chord_embed, dropTokenRate, version_name
        self.total_vf_dim = total_vf_dim
        self.n_experts, self.n_experts_per_token = 6, 2
        self._learned_pos = True
        self._use_rope = version_name in "1.2.3"                     # substring test, as written at :86
        if scene_embed:                          # vf = Linear_vis(features without the scene column) + scene_embedding(offset) (:336-337,481-484)
            self.scene_embedding = nn.Embedding(SCENE_OFFSET_MAX, d_model)
        if chord_embed:
            self.chord_embedding_model = nn.Embedding(CHORD_SIZE, d_model)
            self.chord_embedding_model.weight.requires_grad_(False)
            self._register_load_state_dict_pre_hook(self._resize_chord_table)
        self.embedding = nn.Embedding(CHORD_SIZE, d_model)
        self.embedding_root = nn.Embedding(CHORD_ROOT_SIZE, d_model)
        self.embedding_attr = nn.Embedding(CHORD_ATTR_SIZE, d_model)
        self.Linear_vis = nn.Linear(total_vf_dim, d_model)
        self.Linear_chord = nn.Linear(d_model + 1, d_model)
        self.positional_embedding = nn.Embedding(max_sequence_chord, d_model)
        self.positional_embedding_video = nn.Embedding(max_sequence_video, d_model)
        self.condition_linear = nn.Linear(1, d_model)

        def expert():
            if version_name in ("1.1", "1.3"):
                return GLUExpert(d_model, dim_feedforward, dropout)
            return SiLUExpert(d_model, 2 * d_model, dropout)

        def ff(i):
            if shallow and i < 3:
                return GLUExpert(d_model, dim_feedforward, dropout)
            if version_name in ("1.0", "1.1", "1.3.4"):
                return MoELayer(expert(), d_model, self.n_experts, self.n_experts_per_token, dropout)
            return SharedMoELayer(expert(), d_model, n_experts=self.n_experts, n_experts_per_token=self.n_experts_per_token,
                                  balancing=False, dropout=dropout)

        self.transformer = _TransformerParamsV2(d_model, num_heads, max(3, n_layers) if shallow else n_layers, ff,      # (:114-119)
                                                norm=RMSNorm if rms_norm else nn.LayerNorm)
        self.Wout = nn.Linear(d_model, CHORD_SIZE)
        self.softmax = nn.Softmax(dim=-1)
        if self._use_rope:
            from .rotate_operation import RotaryPositionalEmbeddings
            rope = RotaryPositionalEmbeddings(d_model, max_sequence_video)
            self.register_buffer("_rope_cache", rope.cache.clone(), persistent=False)
        else:
            self._rope_cache = None
        # the RoPE cache caps the chord sequence at max_sequence_video, the positional table at max_sequence_chord
        self._max_dec = min(max_sequence_video, max_sequence_chord) if self._use_rope else max_sequence_chord
        self._derived_sig = None
