"""locomanipulationrl_amd/tasks/base (MI355X loco-manipulation step engine)."""
