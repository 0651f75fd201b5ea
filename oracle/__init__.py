"""CPU oracle for the SA hot path -- TEST INFRASTRUCTURE ONLY (see sa_oracle.c).

Nothing under spsnet_amd/ may import this package.
"""
