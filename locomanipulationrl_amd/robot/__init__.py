"""locomanipulationrl_amd/robot (MI355X loco-manipulation step engine)."""
