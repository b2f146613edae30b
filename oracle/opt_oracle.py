"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the `--tpg OPT` text-prior generator: the reference's None-ResNet-None-CTC recogniser
`crnn.Model(opt)` (model/crnn/model.py:25-110 with the option set of main.py:60-75; feature extractor
model/crnn/modules/feature_extraction.py:54-246: ResNet(BasicBlock, [1, 2, 5, 3]) of FAN), as a function of a parameter dict keyed
like the reference's state_dict.  Pinned: tests/golden/make_golden_next.py imports the genuine reference and asserts this function
reproduces its outputs and every parameter gradient in train and eval mode before it writes the fixtures
(`next_opt.npz`, `train_c3_opt.npz`).  Never imported by the product package."""
import torch
import torch.nn.functional as F

_LAYERS = (("layer1", 1), ("layer2", 2), ("layer3", 5), ("layer4", 3))       # feature_extraction.py:62 ([1, 2, 5, 3])


def _bn(p, pre, x, training):
    """nn.BatchNorm2d defaults: eps 1e-5, momentum 0.1; running statistics updated in place in training mode"""
    return F.batch_norm(x, p[pre + ".running_mean"], p[pre + ".running_var"], p[pre + ".weight"], p[pre + ".bias"], training, 0.1, 1e-5)


def _basic_block(p, pre, x, training):
    """feature_extraction.py:140-172: conv3x3-bn-relu-conv3x3-bn (+ 1x1 conv-bn downsample of the input when the width changes), add, relu"""
    out = F.relu(_bn(p, pre + ".bn1", F.conv2d(x, p[pre + ".conv1.weight"], None, 1, 1), training))
    out = _bn(p, pre + ".bn2", F.conv2d(out, p[pre + ".conv2.weight"], None, 1, 1), training)
    res = x
    if pre + ".downsample.0.weight" in p:
        res = _bn(p, pre + ".downsample.1", F.conv2d(x, p[pre + ".downsample.0.weight"]), training)
    return F.relu(out + res)


def resnet_features(p, x, training, c="FeatureExtraction.ConvNet."):
    """feature_extraction.py:175-246 (ResNet.forward): (N, 1, 32, 100) -> (N, 512, 1, 26)"""
    def cbr(x, conv, bn, k=3, stride=1, pad=1):
        return F.relu(_bn(p, c + bn, F.conv2d(x, p[c + conv + ".weight"], None, stride, pad), training))

    x = cbr(x, "conv0_1", "bn0_1")
    x = cbr(x, "conv0_2", "bn0_2")
    x = F.max_pool2d(x, 2, 2, 0)
    for i in range(_LAYERS[0][1]):
        x = _basic_block(p, f"{c}layer1.{i}", x, training)
    x = cbr(x, "conv1", "bn1")
    x = F.max_pool2d(x, 2, 2, 0)
    for i in range(_LAYERS[1][1]):
        x = _basic_block(p, f"{c}layer2.{i}", x, training)
    x = cbr(x, "conv2", "bn2")
    x = F.max_pool2d(x, 2, (2, 1), (0, 1))
    for i in range(_LAYERS[2][1]):
        x = _basic_block(p, f"{c}layer3.{i}", x, training)
    x = cbr(x, "conv3", "bn3")
    for i in range(_LAYERS[3][1]):
        x = _basic_block(p, f"{c}layer4.{i}", x, training)
    x = cbr(x, "conv4_1", "bn4_1", stride=(2, 1), pad=(0, 1))
    x = cbr(x, "conv4_2", "bn4_2", stride=1, pad=0)
    return x


def opt_forward(p, gray, training=True, **_unused):
    """model.py:81-110: features -> permute(0, 3, 1, 2) -> AdaptiveAvgPool2d((None, 1)) (= mean over the height axis) -> squeeze ->
    Linear(512, num_class) -> permute(1, 0, 2): (N, 1, 32, 100) -> (T = 26, N, num_class), seq-first like CRNN"""
    f = resnet_features(p, gray, training)                    # (N, C, h, w)
    seq = f.permute(0, 3, 1, 2).mean(3)                       # (N, w, C)
    pred = F.linear(seq, p["Prediction.weight"], p["Prediction.bias"])
    return pred.permute(1, 0, 2)
