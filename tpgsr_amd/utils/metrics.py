"""String metrics of the evaluation path (reference: utils/metrics.py:71-88 get_string_crnn, utils/util.py:12-24 str_filt).
The arg-max + CTC collapse runs on the device (tpgsr_ctc_greedy_decode); only N short index lists cross PCIe to become
Python strings."""
import string

import torch

ALPHABET = "-0123456789abcdefghijklmnopqrstuvwxyz"


def ctc_greedy(outputs_: torch.Tensor):
    """outputs_ (T, N, C) logits or probabilities (the recogniser's seq-first output) -> (labels (N, T) int32 padded with -1,
    lengths (N,) int32), both on the device"""
    from .. import kernels as K
    if not outputs_.is_cuda and not K.DRYRUN:
        raise RuntimeError("tpgsr_amd.utils.metrics runs on the GPU only (no CPU fallback)")
    T, N, C = outputs_.shape
    x = outputs_.detach().permute(1, 0, 2)
    x = x if x.is_contiguous() else x.contiguous()           # the drop-in CRNN returns a view of a batch-major buffer: no copy
    labels = torch.empty(N, T, dtype=torch.int32, device=x.device)
    lengths = torch.empty(N, dtype=torch.int32, device=x.device)
    K.ctc_greedy_decode(x.float(), N, T, C, labels, lengths)
    return labels, lengths


def get_string_crnn(outputs_, alphabet=ALPHABET):
    labels, lengths = ctc_greedy(outputs_)
    lab, ln = labels.cpu().tolist(), lengths.cpu().tolist()
    return ["".join(alphabet[i] for i in row[:n]) for row, n in zip(lab, ln)]


def str_filt(str_, voc_type):
    alpha_dict = {"digit": string.digits, "lower": string.digits + string.ascii_lowercase,
                  "upper": string.digits + string.ascii_letters, "all": string.digits + string.ascii_letters + string.punctuation}
    if voc_type == "lower":
        str_ = str_.lower()
    for char in str_:
        if char not in alpha_dict[voc_type]:
            str_ = str_.replace(char, "")
    return str_


def get_vocabulary(voc_type="all", EOS="EOS", PADDING="PADDING", UNKNOWN="UNKNOWN"):
    """utils/labelmaps.py:6-28 (the ASTER recognizer's label set)"""
    import string
    voc = {"digit": string.digits, "lower": string.digits + string.ascii_lowercase, "upper": string.digits + string.ascii_letters,
           "all": string.digits + string.ascii_letters + string.punctuation}[voc_type]
    return list(voc) + [EOS, PADDING, UNKNOWN]


def get_string_aster(output, voc=None, EOS="EOS", UNKNOWN="UNKNOWN"):
    """prediction half of utils/metrics.py:20-70: (N, max_len) label ids -> strings up to the first EOS, UNKNOWN skipped"""
    voc = voc or get_vocabulary("all")
    eos, unk = voc.index(EOS), voc.index(UNKNOWN)
    out = []
    for row in output.tolist():
        chars = []
        for c in row:
            if c == eos:
                break
            if c != unk:
                chars.append(voc[c])
        out.append("".join(chars))
    return out


MORAN_ALPHABET = list(string.digits + string.ascii_lowercase + "$")      # interfaces/base.py:589, :232-234 ('$' ends a word)


def moran_decode(ids, length, alphabet=None):
    """strLabelConverterForAttention.decode (utils/utils_moran.py:79-107): flat label ids + per-sample lengths -> strings"""
    abc = alphabet or MORAN_ALPHABET
    ids = [int(v) for v in (ids.tolist() if hasattr(ids, "tolist") else ids)]
    lens = [int(v) for v in (length.tolist() if hasattr(length, "tolist") else length)]
    if len(ids) != sum(lens):
        raise ValueError(f"texts with length: {len(ids)} does not match declared length: {sum(lens)}")
    out, i = [], 0
    for n in lens:
        out.append("".join(abc[c] for c in ids[i:i + n]))
        i += n
    return out


def get_string_moran(preds, length, alphabet=None):
    """interfaces/super_resolution.py:1393-1396: arg-max of the (sum(length), nclass) class scores of MORAN's left-to-right decoder,
    decoded, every word cut at its first '$'"""
    if preds.is_cuda:       # first arg-max per row on the device (tpgsr_softmax_max), like the decoder's own feedback
        from .. import kernels as K
        rows, ncls = preds.shape
        ids = torch.empty(rows, dtype=torch.int32, device=preds.device)
        K.softmax_max(preds.contiguous().float(), rows, ncls, ids, torch.empty(rows, device=preds.device), 1, 0)
    else:
        ids = preds.argmax(1)
    return [s.split("$")[0] for s in moran_decode(ids, length, alphabet)]
