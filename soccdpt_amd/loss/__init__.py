"""Freeze / unfreeze helpers of the reference's training scripts (/root/reference/SOccDPT/loss/__init__.py:15-50), same names and the same
INDEX-based semantics: `unfreeze_pretrained_encoder_by_percentage` makes the FIRST round(N * percentage) parameters of
`model.pretrained.parameters()` trainable (registration order, patch embedding first) and freezes the rest."""


def freeze_pretrained_encoder(model):
    for param in model.pretrained.parameters():
        param.requires_grad = False


def _unfreeze_first(parameters, percentage):
    assert 0 <= percentage <= 1, "percentage must be between 0 and 1"
    parameters = list(parameters)
    m = round(len(parameters) * percentage)
    for index, param in enumerate(parameters):
        param.requires_grad = index < m


def unfreeze_pretrained_encoder_by_percentage(model, percentage):
    _unfreeze_first(model.pretrained.parameters(), percentage)


def unfreeze_module_by_percentage(model, percentage):
    _unfreeze_first(model.parameters(), percentage)


def unfreeze_module(model):
    for param in model.parameters():
        param.requires_grad = True
