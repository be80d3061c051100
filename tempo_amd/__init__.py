"""tempo_amd: MI355X-native execution engine for tempo's TSDF hot path."""
