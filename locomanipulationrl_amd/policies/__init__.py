"""locomanipulationrl_amd/policies (MI355X loco-manipulation step engine)."""
