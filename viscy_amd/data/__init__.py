from .hcs import HCSDataModule, SlidingWindowDataset  # noqa: F401
from .ome_zarr import open_ome_zarr, write_hcs_plate  # noqa: F401
from .combined import (BatchedConcatDataModule, CombinedDataModule, CombinedLoader, CombineMode, ConcatDataModule,  # noqa: F401
                       ShardedDistributedSampler)
