"""torch.hub entry point: torch.hub.load("/path/to/this/repo", "mvtracker", source="local", checkpoint=...)."""
dependencies = ["torch", "numpy"]


def mvtracker(pretrained: bool = False, device="cuda", checkpoint=None, **kwargs):
    from mvtracker_amd.factory import load_mvtracker
    return load_mvtracker(checkpoint=checkpoint, device=device, pretrained=pretrained, **kwargs)
