"""locomanipulationrl_amd/utils/domain_randomization (MI355X loco-manipulation step engine)."""
