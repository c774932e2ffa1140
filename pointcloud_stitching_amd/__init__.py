"""pointcloud_stitching_amd — MI355X-native deproject -> transform -> pack hot path."""
