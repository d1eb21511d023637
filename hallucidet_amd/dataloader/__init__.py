from .dataloader import (DevicePrefetcher, MultiModalDataModule, MultiModalDetectionDataset, SingleModalDataModule,  # noqa: F401
                         SingleModalDetectionDataset, get_bbox, split_dataset)
