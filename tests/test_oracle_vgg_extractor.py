"""The oracle's restatement of the VGG19 feature taps (oracle/model_ref.py:content_loss; reference model.py:289-335) against a
torch.fx feature extractor built the way torchvision's `create_feature_extractor` builds one: trace the module, make the
selected nodes the graph's output, prune everything behind the last one, recompile.  torchvision itself is absent from this
image, so the traced module is a plain-torch clone of `vgg19().features` (cfg "E": 3x3 convs + ReLU(inplace=True) + max-pools,
the same module indices); the weights are random -- the pretrained numbers stay unpinned, the GRAPH semantics do not:
with inplace ReLUs every tapped conv output except the last has been overwritten by its ReLU when the dict is returned."""
import torch
import torch.fx
from torch import nn

from oracle import model_ref as M

NODES = ["features.2", "features.7", "features.16", "features.25", "features.34"]      # reference config.py:131


class VGGLike(nn.Module):
    def __init__(self, inplace: bool) -> None:
        super().__init__()
        layers, cin = [], 3
        for v in M.VGG19_CFG + ["M"]:                       # torchvision's cfg "E" ends with a pool (features.36)
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=inplace)]
                cin = v
        self.features = nn.Sequential(*layers)

    def forward(self, x):
        return self.features(x)


def make_extractor(model: nn.Module, return_nodes):
    gm = torch.fx.symbolic_trace(model)
    by_target = {str(n.target): n for n in gm.graph.nodes if n.op == "call_module"}
    outs = {k: by_target[k] for k in return_nodes}
    for n in list(gm.graph.nodes):
        if n.op == "output":
            gm.graph.erase_node(n)
    gm.graph.output(outs)
    gm.graph.eliminate_dead_code()                          # drops features.35 / features.36: nothing behind the last tap runs
    gm.recompile()
    return gm


def test_oracle_feature_taps_equal_an_fx_extractor_with_inplace_relus():
    torch.manual_seed(3)
    model = VGGLike(inplace=True).eval()
    assert len(model.features) == 37 and all(isinstance(model.features[int(k.split(".")[1])], nn.Conv2d) for k in NODES)
    ext = make_extractor(model, NODES)
    assert "features_35" not in {n.name for n in ext.graph.nodes}
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]      # reference config.py:132-133
    sr, hr = torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32)
    norm = lambda t: (t - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)   # noqa: E731
    with torch.no_grad():
        fa, fb = ext(norm(sr)), ext(norm(hr))
        want = tuple(torch.nn.functional.l1_loss(fa[k], fb[k]) for k in NODES)
        got_alias = M.content_loss(sr, hr, sd, NODES, mean, std, inplace_relu_aliasing=True)
        got_plain = M.content_loss(sr, hr, sd, NODES, mean, std, inplace_relu_aliasing=False)
    for k in NODES[:-1]:
        assert float(fa[k].min()) >= 0.0, k                 # the returned "conv output" is the ReLU-ed tensor
    assert float(fa[NODES[-1]].min()) < 0.0                 # the last one is not: its ReLU was pruned
    for w, a in zip(want, got_alias):
        assert abs(w.item() - a.item()) <= 1e-6 * max(1.0, abs(w.item()))
    assert any(abs(w.item() - p.item()) > 1e-3 * abs(w.item()) for w, p in zip(want[:-1], got_plain[:-1]))
    # without inplace ReLUs the same extractor returns the pre-activation tensors -- the switch's other position
    ext2 = make_extractor(VGGLike(inplace=False).eval(), NODES)
    ext2.load_state_dict(model.state_dict())
    with torch.no_grad():
        fa2, fb2 = ext2(norm(sr)), ext2(norm(hr))
    for k, p in zip(NODES, got_plain):
        assert abs(torch.nn.functional.l1_loss(fa2[k], fb2[k]).item() - p.item()) <= 1e-6 * max(1.0, abs(p.item()))
