"""Decoders with the reference's module tree and state_dict keys; compute runs in the HIP library.

Mirror of src/conv_onet/models/decoder.py (reference): GaussianFourierFeatureTransform :7-30,
DenseLayer :70-79, MLP :91-203, MLP_no_xyz :206-274, NICE :277-342.  The modules only OWN parameters
(`pts_linears.{i}.{weight,bias}`, `fc_c.{i}.{weight,bias}`, `embedder._B`, `output_linear.{weight,bias}`)
so that load_state_dict / deepcopy / share_memory / optimizers work as with the reference;
`NICE.forward(p, c_grid, stage)` evaluates through enslam_eval_points (forward only, like its
callers Mesher.py:308 and Renderer.eval_points)."""
import torch
import torch.nn as nn

from . import functional as EF


class GaussianFourierFeatureTransform(nn.Module):
    """sin(x @ B); B [3, mapping_size] learnable, randn * scale at init (decoder.py:17-30)."""

    def __init__(self, num_input_channels, mapping_size=93, scale=25, learnable=True):
        super().__init__()
        b = torch.randn((num_input_channels, mapping_size)) * scale
        if learnable:
            self._B = nn.Parameter(b)
        else:
            self.register_buffer('_B', b, persistent=False)


class DenseLayer(nn.Linear):
    """nn.Linear with Xavier-uniform(gain(activation)) weights and zero bias (decoder.py:70-79)."""

    def __init__(self, in_dim, out_dim, activation="relu", *args, **kwargs):
        self.activation = activation
        super().__init__(in_dim, out_dim, *args, **kwargs)

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain(self.activation))
        if self.bias is not None:
            nn.init.zeros_(self.bias)


class MLP(nn.Module):
    """middle / fine / color decoder parameters (decoder.py:110-166); fourier embedding only."""

    def __init__(self, name='', dim=3, c_dim=128, hidden_size=256, n_blocks=5, leaky=False, sample_mode='bilinear',
                 color=False, skips=[2], grid_len=0.16, pos_embedding_method='fourier', concat_feature=False):
        super().__init__()
        if pos_embedding_method != 'fourier' or leaky or sample_mode != 'bilinear':
            raise NotImplementedError("only the shipped NICE configuration (fourier / relu / bilinear) is built")
        if hidden_size != 32 or n_blocks != 5 or list(skips) != [2] or c_dim not in (32, 64):
            raise NotImplementedError("kernels are specialised for hidden 32, 5 blocks, skip at 2, c_dim 32/64")
        self.name, self.color, self.c_dim, self.grid_len = name, color, c_dim, grid_len
        self.concat_feature, self.n_blocks, self.skips = concat_feature, n_blocks, skips
        self.no_grad_feature = False
        self.sample_mode = sample_mode
        embedding_size = 93
        self.fc_c = nn.ModuleList([nn.Linear(c_dim, hidden_size) for _ in range(n_blocks)])
        self.embedder = GaussianFourierFeatureTransform(dim, mapping_size=embedding_size, scale=25)
        self.pts_linears = nn.ModuleList(
            [DenseLayer(embedding_size, hidden_size, activation="relu")] +
            [DenseLayer(hidden_size, hidden_size, activation="relu") if i not in self.skips
             else DenseLayer(hidden_size + embedding_size, hidden_size, activation="relu")
             for i in range(n_blocks - 1)])
        self.output_linear = DenseLayer(hidden_size, 4 if color else 1, activation="linear")


class MLP_no_xyz(nn.Module):
    """coarse decoder parameters (decoder.py:223-252)."""

    def __init__(self, name='', dim=3, c_dim=128, hidden_size=256, n_blocks=5, leaky=False, sample_mode='bilinear',
                 color=False, skips=[2], grid_len=0.16):
        super().__init__()
        if hidden_size != 32 or n_blocks != 5 or list(skips) != [2] or c_dim != 32 or color or leaky:
            raise NotImplementedError("kernels are specialised for hidden 32, 5 blocks, skip at 2, c_dim 32")
        self.name, self.color, self.c_dim, self.grid_len = name, color, c_dim, grid_len
        self.n_blocks, self.skips, self.no_grad_feature, self.sample_mode = n_blocks, skips, False, sample_mode
        self.pts_linears = nn.ModuleList(
            [DenseLayer(hidden_size, hidden_size, activation="relu")] +
            [DenseLayer(hidden_size, hidden_size, activation="relu") if i not in self.skips
             else DenseLayer(hidden_size + c_dim, hidden_size, activation="relu") for i in range(n_blocks - 1)])
        self.output_linear = DenseLayer(hidden_size, 1, activation="linear")


class NICE(nn.Module):
    """Neural Implicit Scalable Encoding: {coarse,middle,fine,color}_decoder (decoder.py:293-310).
    `.bound` (and each sub-decoder's `.bound`) are assigned by the caller as in EvenNICER_SLAM.py:177-182."""

    def __init__(self, dim=3, c_dim=32, coarse_grid_len=2.0, middle_grid_len=0.16, fine_grid_len=0.16,
                 color_grid_len=0.16, hidden_size=32, coarse=False, pos_embedding_method='fourier'):
        super().__init__()
        if coarse:
            self.coarse_decoder = MLP_no_xyz(name='coarse', dim=dim, c_dim=c_dim, color=False,
                                             hidden_size=hidden_size, grid_len=coarse_grid_len)
        self.middle_decoder = MLP(name='middle', dim=dim, c_dim=c_dim, color=False, skips=[2], n_blocks=5,
                                  hidden_size=hidden_size, grid_len=middle_grid_len,
                                  pos_embedding_method=pos_embedding_method)
        self.fine_decoder = MLP(name='fine', dim=dim, c_dim=c_dim * 2, color=False, skips=[2], n_blocks=5,
                                hidden_size=hidden_size, grid_len=fine_grid_len, concat_feature=True,
                                pos_embedding_method=pos_embedding_method)
        self.color_decoder = MLP(name='color', dim=dim, c_dim=c_dim, color=True, skips=[2], n_blocks=5,
                                 hidden_size=hidden_size, grid_len=color_grid_len,
                                 pos_embedding_method=pos_embedding_method)

    def forward(self, p, c_grid, stage='middle', **kwargs):
        """raw [N,4] for points p ([1,N,3] or [N,3]); no bound mask (that is Renderer.eval_points)."""
        bound = getattr(self, 'bound', None)
        if bound is None:
            bound = self.middle_decoder.bound
        return EF.eval_points(p.reshape(-1, 3), self, c_grid, stage, bound, apply_mask=False)


def get_model(cfg, nice=True):
    """conv_onet/config.py:4-33 factory (NICE only; the iMAP MLP is outside the hot path)."""
    if not nice:
        raise NotImplementedError("iMAP decoder is not part of the accelerated path")
    gl = cfg['grid_len']
    return NICE(dim=cfg['data']['dim'], c_dim=cfg['model']['c_dim'], coarse=cfg['coarse'],
                coarse_grid_len=gl['coarse'], middle_grid_len=gl['middle'], fine_grid_len=gl['fine'],
                color_grid_len=gl['color'], pos_embedding_method=cfg['model']['pos_embedding_method'])
