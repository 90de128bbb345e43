from .flow_matching_module import CNF, SetFlowMatchingLitModule, ode_wrapper  # noqa: F401
