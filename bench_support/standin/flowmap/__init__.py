"""Stand-in for the reference package's module layout (bench_support/standin/README.md).  Test infrastructure.

Its own arithmetic — what its functions do while nothing is installed — is the oracle's (oracle/flowmap_oracle.py), reached through the proxy
below: the oracle is imported only when that arithmetic first RUNS, and it refuses device tensors.  So a process that installs flowmap_amd
into this package and steps it on the GPU (tests/test_install_standin.py, bench.py's default `--model installed`) never imports the oracle
and cannot be served by it: a name install() failed to rebind raises here instead of quietly computing with torch.
"""
import torch


class _HostOnlyOracle:
    def __getattr__(self, name):
        from oracle import flowmap_oracle

        attr = getattr(flowmap_oracle, name)
        if not callable(attr) or isinstance(attr, type):
            return attr

        def host_only(*args, **kwargs):
            for value in (*args, *kwargs.values()):
                if torch.is_tensor(value) and value.is_cuda:
                    raise RuntimeError(f"bench_support/standin: the stand-in's own arithmetic ({name}: the oracle's, host-only test infrastructure) was reached "
                                       "with a GPU tensor — flowmap_amd.install() did not rebind the caller")
            return attr(*args, **kwargs)

        return host_only


orc = _HostOnlyOracle()
