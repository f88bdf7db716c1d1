"""TEST INFRASTRUCTURE ONLY - NOT BoTorch.  The reference turns its objectives into ``botorch.acquisition.objective`` modules even
for plain dataframe transformations (``Objective.transform`` -> ``to_botorch``, objectives/base.py:107-110, 225-260 - called by
``simulate_experiment`` for its ``*_IterBest`` columns, simulation/core.py:222-227, and by ``Objective._pre_transform``).  Those
modules are callable wrappers without arithmetic of their own; this package declares the four the reference's target / objective
transformations instantiate, so that the reference's own code paths run in an image without BoTorch.  No model, posterior,
acquisition function, sampler or optimiser exists here: everything numerical in the reference still fails without the real package.
"""

__stub__ = True
__version__ = "0+test-stub"
