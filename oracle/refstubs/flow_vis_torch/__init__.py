"""Import-only stand-in for flow_vis_torch (absent here): the reference's visualisers import `flow_to_color`; nothing of it runs in the tests."""


def flow_to_color(*args, **kwargs):
    raise RuntimeError("flow_vis_torch.flow_to_color: import-only stand-in (oracle/refstubs)")
