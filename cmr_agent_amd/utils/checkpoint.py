"""Checkpoint loading with the reference's strictness (Test_Agent.py:129-136 / Train_Agent.py:104-107 call
load_state_dict strictly).  Two key families may legitimately be absent: BatchNorm `num_batches_tracked` counters
(the closed-form weight fill skips them) and the image-size specific `position_embeddings` table of ImageViT
(ImageViT.py:26-27; recomputed by the module when T differs).  Anything else missing or unexpected is an error, so a
checkpoint with a `module.` prefix, a wrapper dict or renamed keys can never silently evaluate default weights."""

OPTIONAL_SUFFIXES = ("num_batches_tracked", "position_embeddings")


def load_checked(module, state_dict):
    if not isinstance(state_dict, dict) or not state_dict:
        raise ValueError("load_checked: expected a non-empty state_dict, got %s" % type(state_dict).__name__)
    missing, unexpected = module.load_state_dict(state_dict, strict=False)
    bad_missing = [k for k in missing if not k.endswith(OPTIONAL_SUFFIXES)]
    if unexpected or bad_missing:
        raise RuntimeError("%s: checkpoint does not match the module (strict load, like the reference): "
                           "%d unexpected keys (first: %s), %d missing keys (first: %s)" % (
                               type(module).__name__, len(unexpected), list(unexpected)[:3], len(bad_missing), bad_missing[:3]))
    return module
